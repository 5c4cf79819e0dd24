#!/bin/bash
# round-2 GPU session Y: single-transcendental GELU in the GeGLU / BERT MLP epilogues -- parity, GEMM rates, bench
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2y; mkdir -p $O
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_bert_gpu.py tests/test_heads_gpu.py tests/test_full_shapes_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_gpu.txt
timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep "geglu\|none   M=65536 N=2304" | tee $O/gemm.txt
timeout 300 python bench.py --cpu-budget 0 --no-profile --steps 12 2>/dev/null | tail -1 | cut -c100-200 | tee $O/bench.txt
