REPO=$(pwd); OUT=$REPO/gpurun_out/r6_misc_timeline; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in dense1 sparse1 sparse16; do
WHAT=$w rocprofv3 --kernel-trace -d $OUT/kt_$w -- python $REPO/tools/probes/misc_timeline.py > $OUT/kt_$w.log 2>&1
python - "$(find $OUT/kt_$w -name '*.db' | head -1)" > $OUT/timeline_$w.txt 2>&1 <<'PY'
import sqlite3, sys
sys.path.insert(0, "/root/repo/tools")
from kname import pretty
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, duration, grid_x from kernels order by start").fetchall()
rows = rows[-28:]
t0 = rows[0][1]
for name, start, dur, gx in rows:
    print(f"{(start - t0) / 1e3:9.1f} us  +{dur / 1e3:8.1f} us  grid {gx:8d}  {pretty(name)[:100]}")
PY
rm -rf $OUT/kt_$w
echo "== $w"; cat $OUT/timeline_$w.txt
done
