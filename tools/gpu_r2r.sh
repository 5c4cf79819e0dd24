#!/bin/bash
# round-2 GPU session R: do the epilogue store bursts of the 256 CUs coincide?  start-time stagger probe
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2r; mkdir -p $O
for st in 0 64 128 256 0; do
  echo "== VRAG_GEMM_STAGGER=$st"
  VRAG_GEMM_STAGGER=$st timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep -v "amdgpu.ids\|none"
done | tee $O/stagger.txt
for st in 0 128; do
VRAG_GEMM_STAGGER=$st timeout 300 python bench.py --cpu-budget 0 --no-profile --steps 10 2>/dev/null | tail -1 | cut -c100-200 | tee -a $O/stagger.txt
done
