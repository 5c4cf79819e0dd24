REPO=$(pwd); OUT=$REPO/gpurun_out/r6_tiled_trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for NQ in 64 256; do
NQ=$NQ rocprofv3 --kernel-trace --stats -d $OUT/kt$NQ -- python $REPO/tools/probes/tiled_trace.py > $OUT/kt$NQ.log 2>&1
python $REPO/tools/rocpd_summary.py "$(find $OUT/kt$NQ -name '*.db' | head -1)" > $OUT/stats$NQ.txt 2>&1
rm -rf $OUT/kt$NQ
done
head -30 $OUT/stats64.txt | cut -c1-200; head -20 $OUT/stats256.txt | cut -c1-200
