REPO=$(pwd); OUT=$REPO/gpurun_out/r6_embed_latency; mkdir -p $OUT
python tools/probes/embed_latency_probe.py 2>/dev/null | grep '^{' | tee $OUT/phases.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $REPO/tools/probes/embed_latency_probe.py > $OUT/kt.log 2>&1
python $REPO/tools/rocpd_summary.py "$(find $OUT/kt -name '*.db' | head -1)" > $OUT/stats.txt 2>&1; rm -rf $OUT/kt
head -22 $OUT/stats.txt | cut -c1-200
