#!/bin/bash
# Evidence for the rows next to the headline path (run through gpurun from the repo root):
#   tools/profile_aux.sh r01   -> gpurun_out/<tag>_aux/{topk_lines.json, topk_kernel_stats.txt, embed_lines.json, embed_kernel_stats.txt}
TAG=${1:-round}
REPO=$(pwd)
OUT=$REPO/gpurun_out/${TAG}_aux
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt_topk" -- python $REPO/tools/bench_topk.py 2>/dev/null | grep '^{' > "$OUT/topk_lines.json"
rocprofv3 --kernel-trace --stats -d "$OUT/kt_embed" -- python $REPO/tools/bench_embed.py 2>/dev/null | grep '^{' > "$OUT/embed_lines.json"
rocprofv3 --kernel-trace --stats -d "$OUT/kt_f32" -- python $REPO/tools/bench_dense_f32.py 2>/dev/null | grep '^{' > "$OUT/dense_f32_lines.json"
# counters of the top-k kernels (own run, no trace flags): matrix-pipe busy cycles of the tiled dense search's GEMM, LDS conflicts
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  -d "$OUT/pmc_topk" -- python $REPO/tools/bench_topk.py > "$OUT/pmc_topk.log" 2>&1
cd "$REPO"
python tools/pmc_summary.py "$(find $OUT/pmc_topk -name '*.db' | head -1)" > "$OUT/topk_pmc.txt" 2>&1
rm -rf "$OUT/pmc_topk"
python tools/rocpd_summary.py "$(find $OUT/kt_topk -name '*.db' | head -1)" > "$OUT/topk_kernel_stats.txt" 2>&1
python tools/rocpd_summary.py "$(find $OUT/kt_embed -name '*.db' | head -1)" > "$OUT/embed_kernel_stats.txt" 2>&1
python tools/rocpd_summary.py "$(find $OUT/kt_f32 -name '*.db' | head -1)" > "$OUT/dense_f32_kernel_stats.txt" 2>&1
rm -rf "$OUT/kt_topk" "$OUT/kt_embed" "$OUT/kt_f32"
head -12 "$OUT/topk_kernel_stats.txt" | cut -c1-180; head -12 "$OUT/embed_kernel_stats.txt" | cut -c1-180
