#!/bin/bash
# round-2 GPU session D: fp16 operand mode + split-operand token head; full suite; bf16 vs f16 bench
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2d; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 300 python tools/gemm_bench.py 65536 > $O/gemm_bf16.log 2>&1
VRAG_DEBUG_GEMM_F16=1 timeout 300 python tools/gemm_bench.py 65536 > $O/gemm_f16.log 2>&1
timeout 600 python bench.py --cpu-budget 0 > $O/bench_bf16.json 2> $O/bench_bf16.err
tail -25 $O/pytest.log; paste $O/gemm_bf16.log $O/gemm_f16.log | cut -c1-200; python -c "import json; d=json.loads(open('$O/bench_bf16.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
