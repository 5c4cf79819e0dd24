#!/bin/bash
# The top-k part of tools/profile_aux.sh alone (kernel traces of bench_topk.py / bench_dense_f32.py + the SQ counter pass):
#   tools/profile_topk.sh r05e -> gpurun_out/<tag>_topk/{topk_lines.json, topk_kernel_stats.txt, topk_pmc.txt, dense_f32_lines.json, dense_f32_kernel_stats.txt}
TAG=${1:-round}
REPO=$(pwd)
OUT=$REPO/gpurun_out/${TAG}_topk
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt_topk" -- python $REPO/tools/bench_topk.py 2>/dev/null | grep '^{' > "$OUT/topk_lines.json"
rocprofv3 --kernel-trace --stats -d "$OUT/kt_f32" -- python $REPO/tools/bench_dense_f32.py 2>/dev/null | grep '^{' > "$OUT/dense_f32_lines.json"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  -d "$OUT/pmc_topk" -- python $REPO/tools/bench_topk.py > "$OUT/pmc_topk.log" 2>&1
cd "$REPO"
python tools/pmc_summary.py "$(find $OUT/pmc_topk -name '*.db' | head -1)" > "$OUT/topk_pmc.txt" 2>&1
python tools/rocpd_summary.py "$(find $OUT/kt_topk -name '*.db' | head -1)" > "$OUT/topk_kernel_stats.txt" 2>&1
python tools/rocpd_summary.py "$(find $OUT/kt_f32 -name '*.db' | head -1)" > "$OUT/dense_f32_kernel_stats.txt" 2>&1
rm -rf "$OUT/pmc_topk" "$OUT/kt_topk" "$OUT/kt_f32"
head -8 "$OUT/topk_kernel_stats.txt" | cut -c1-180
