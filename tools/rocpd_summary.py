#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd SQLite result (--kernel-trace --stats) into the per-kernel summary
text committed under profiles/.  Usage: tools/rocpd_summary.py results.db [> profiles/x.txt]"""
import sqlite3
import sys


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
              f"{r[10] or 0:5d} {r[11] or 0:9d} {r[12] or 0:4d}  {r[0]}")


if __name__ == "__main__":
    main(sys.argv[1])
