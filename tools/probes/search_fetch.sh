# HBM fetch bytes of the tiled dense search's kernels (own run, no trace flags): FETCH_SIZE per dispatch, summed per kernel by tools/pmc_summary.py
REPO=$(pwd); OUT=$REPO/gpurun_out/r6_search_fetch; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MODE=bf16 NQ=64 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc -- python $REPO/tools/probes/search_timeline.py > $OUT/pmc.log 2>&1
cd $REPO
python tools/pmc_summary.py "$(find $OUT/pmc -name '*.db' | head -1)" > $OUT/fetch_bf16_64.txt 2>&1
rm -rf $OUT/pmc
head -12 $OUT/fetch_bf16_64.txt | cut -c1-220
