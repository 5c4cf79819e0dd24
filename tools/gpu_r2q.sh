#!/bin/bash
# round-2 GPU session Q: unswitched epilogues (QKV / GeGLU / bf16) -- GEMM rates, parity subset, bench
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2q; mkdir -p $O
timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep -v amdgpu.ids | tee $O/gemm_bench.txt
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_bert_gpu.py tests/test_heads_gpu.py tests/test_full_shapes_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
timeout 300 python bench.py --cpu-budget 0 --no-profile --steps 10 2>/dev/null | tail -1 | cut -c1-260 | tee $O/bench.json
