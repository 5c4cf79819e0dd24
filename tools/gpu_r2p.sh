#!/bin/bash
# round-2 GPU session P: GEMM template on v_mfma 16x16x32 -- parity suite, GEMM rates, bench line
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2p; mkdir -p $O
timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep -v amdgpu.ids | tee $O/gemm_bench.txt
timeout 300 python tools/gemm_bench.py cal 50 2>&1 | grep -v amdgpu.ids | tee -a $O/gemm_bench.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 1500 $O/bench_line.json
timeout 300 python bench.py --cpu-budget 0 --no-profile --steps 8 --operand-dtype f16 2>/dev/null | tail -1 | cut -c1-260 | tee $O/bench_f16.json
