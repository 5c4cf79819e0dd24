"""Prints the dispatches of the LAST search of a rocprofv3 kernel trace of search_timeline.py: start offset, duration, grid, name."""
import os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kname import pretty
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, duration, grid_x from kernels order by start").fetchall()
# the last search starts at the last tiled_queries_kernel (or the last prefilter launch chain: take the last 40 dispatches otherwise)
idx = [i for i, r in enumerate(rows) if "tiled_queries" in r[0]]
lo = idx[-1] if idx else max(0, len(rows) - 40)
t0 = rows[lo][1]
for name, start, dur, gx in rows[lo:]:
    print(f"{(start - t0) / 1e3:9.1f} us  +{dur / 1e3:8.1f} us  grid {gx:8d}  {pretty(name)[:110]}")
print(f"span {(rows[-1][1] + rows[-1][2] - t0) / 1e3:.1f} us")
