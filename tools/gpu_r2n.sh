#!/bin/bash
# round-2 GPU session N: 4-wave / 128x128-per-wave GEMM layout (VRAG_GEMM_W4): main-loop rate, epilogue kernels, parity
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2n; mkdir -p $O
for w in 0 1 2; do
  echo "== VRAG_GEMM_W4=$w"
  VRAG_GEMM_W4=$w timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep -v amdgpu.ids
  VRAG_GEMM_W4=$w timeout 300 python tools/gemm_bench.py cal 50 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $O/gemm_bench.txt
VRAG_GEMM_W4=2 timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_heads_gpu.py tests/test_full_shapes_gpu.py -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_w4.txt
VRAG_GEMM_W4=2 timeout 300 python bench.py --cpu-budget 0 --no-profile --steps 8 2>/dev/null | tail -1 | cut -c1-300 | tee $O/bench_w4.json
