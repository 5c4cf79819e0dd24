#!/bin/bash
# round-2 GPU session AF: chained MFMA order in every GEMM -- full GPU suite, GEMM rates, bench A/B is against the numbers of session AD/AE
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2af; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $O/pytest.txt
timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep -v amdgpu | tee $O/gemm.txt
for i in 1 2; do timeout 300 python bench.py --cpu-budget 0 --no-profile --steps 12 2>/dev/null | tail -1 | cut -c100-200; done | tee $O/bench.txt
