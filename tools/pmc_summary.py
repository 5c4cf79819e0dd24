#!/usr/bin/env python3
"""Summarises a rocprofv3 --pmc rocpd database: per kernel name, mean counter value per dispatch."""
import os
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import pretty  # noqa: E402


def main(path):
    db = sqlite3.connect(path)
    cols = [c[1] for c in db.execute("pragma table_info('counters_collection')")]
    # columns of interest vary by version; find name/counter/value columns
    name_c = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
    cn = "counter_name" if "counter_name" in cols else None
    val = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
    did = "dispatch_id" if "dispatch_id" in cols else None
    if not (name_c and cn and val):
        print("columns:", cols)
        return
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
    for k, c, v, d in db.execute(f"select {name_c}, {cn}, {val}, {did or 0} from counters_collection"):
        a = agg[pretty(k)][c]
        a[0] += float(v)
        a[1].add(d)
    for k, cs in agg.items():
        print(k)
        for c, (tot, ds) in sorted(cs.items()):
            n = max(1, len(ds))
            print(f"   {c:34s} {tot / n:18.1f} per dispatch  ({n} dispatches)")


if __name__ == "__main__":
    main(sys.argv[1])
