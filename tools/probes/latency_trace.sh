# Kernel trace of the per-query call (tools/bench_extract_latency.py): per-kernel device times at ~1 000 packed rows
REPO=$(pwd); OUT=$REPO/gpurun_out/r6_latency_trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $REPO/tools/bench_extract_latency.py > $OUT/kt.log 2>&1
python $REPO/tools/rocpd_summary.py "$(find $OUT/kt -name '*.db' | head -1)" > $OUT/stats.txt 2>&1
rm -rf $OUT/kt
grep '^{' $OUT/kt.log; head -24 $OUT/stats.txt | cut -c1-220
