#!/bin/bash
# round-2 GPU session AB: split residual stream (two 16-bit planes + per-row shift instead of fp32 rows + operand copy)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2ab; mkdir -p $O
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_heads_gpu.py tests/test_full_shapes_gpu.py tests/test_fuzz_gpu.py tests/test_extractor_gpu.py tests/test_bert_gpu.py -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep "resid" | tee $O/gemm.txt
VRAG_DEBUG_GEMM_F32_STREAM=1 timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep "resid" | sed 's/^/f32-stream /' | tee -a $O/gemm.txt
for v in 1 0 1 0; do
VRAG_SPLIT_STREAM=$v timeout 300 python bench.py --cpu-budget 0 --no-profile --steps 12 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('split', $v, round(d['value'],1), round(d['ms_per_step'],3), d.get('parity_max_abs_err_vs_oracle'))"
done | tee $O/bench.txt
