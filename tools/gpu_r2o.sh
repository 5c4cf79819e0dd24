#!/bin/bash
# round-2 GPU session O: MFMA shape probe -- 4-wave main loop with 32x32x16 (W4=2) vs 16x16x32 (W4=3, garbage results by design)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2o; mkdir -p $O
for w in 0 4 0 4 3; do
  echo "== VRAG_GEMM_W4=$w"
  VRAG_GEMM_W4=$w timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep "none"
  VRAG_GEMM_W4=$w timeout 300 python tools/gemm_bench.py cal 50 2>&1 | grep "none"
done 2>&1 | tee $O/gemm_bench.txt
