#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd SQLite result (--kernel-trace --stats) into the per-kernel summary
text committed under profiles/.  Usage: tools/rocpd_summary.py results.db [W K] [> profiles/x.txt]

With W K (the --warmup/--steps the profiled `bench.py` ran with) two more tables are printed: the
dispatches of the timed region (micro-batches on two streams) and those of bench.py's single-stream
pass (1 untimed + 2 profiled steps, always last) -- the phases bench.py's `roofline.timed_region`
and `roofline` report with HIP events."""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import pretty  # noqa: E402


def main(path: str) -> None:
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print(f"# total kernel time: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6} "
          f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>7} {'scr':>5} {'grid_x':>9} {'wg':>4}  name")
    for r in rows:
        print(f"{r[1]:7d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.2f} {r[4] / 1e3:9.2f} {r[5] / 1e3:9.2f} "
              f"{100.0 * r[2] / total:6.2f} {r[6] or 0:5d} {r[7] or 0:5d} {r[8] or 0:5d} {r[9] or 0:7d} "
              f"{r[10] or 0:5d} {r[11] or 0:9d} {r[12] or 0:4d}  {pretty(r[0])}")


def phases(path: str, W: int, K: int) -> None:
    db = sqlite3.connect(path)
    per = {}
    for name, start, dur in db.execute("select name, start, duration from kernels order by start"):
        per.setdefault(pretty(name), []).append(dur)
    tot = W + K + 3
    for title, lo, hi in (("timed region (two streams)", W, W + K), ("single-stream pass (profiled steps)", W + K + 1, tot)):
        print(f"\n# {title}: dispatches of bench steps [{lo}, {hi}) of {tot}")
        print(f"{'calls':>7} {'avg_us':>10} {'min_us':>9} {'max_us':>9}  name")
        for name, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            if len(d) % tot or "vrag::" not in name or "cvt_rows" in name:
                continue
            L = len(d) // tot
            seg = d[lo * L:hi * L]
            print(f"{len(seg):7d} {sum(seg) / len(seg) / 1e3:10.2f} {min(seg) / 1e3:9.2f} {max(seg) / 1e3:9.2f}  {name}")


if __name__ == "__main__":
    main(sys.argv[1])
    if len(sys.argv) >= 4:
        phases(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
