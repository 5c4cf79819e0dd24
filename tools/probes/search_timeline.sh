REPO=$(pwd); OUT=$REPO/gpurun_out/r6_search_timeline; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "bf16 64" "f32 32" "bf16 256"; do
set -- $cfg
MODE=$1 NQ=$2 rocprofv3 --kernel-trace -d $OUT/kt_$1_$2 -- python $REPO/tools/probes/search_timeline.py > $OUT/kt_$1_$2.log 2>&1
python $REPO/tools/probes/search_timeline_dump.py "$(find $OUT/kt_$1_$2 -name '*.db' | head -1)" > $OUT/timeline_$1_$2.txt 2>&1
rm -rf $OUT/kt_$1_$2
done
cat $OUT/timeline_*.txt
