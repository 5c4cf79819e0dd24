#!/bin/bash
# round-2 GPU session B: residual-GEMM pair configuration (2 workgroups / CU) A/B + parity
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
for v in 1 0; do
  VRAG_GEMM_RES_PAIR=$v timeout 300 python tools/gemm_bench.py 65536 > $O/gemm_shapes_pair$v.log 2>&1
  VRAG_GEMM_RES_PAIR=$v timeout 600 python bench.py --cpu-budget 0 > $O/bench_pair$v.json 2> $O/bench_pair$v.err
  VRAG_GEMM_RES_PAIR=$v VRAG_STREAMS=1 timeout 600 python bench.py --cpu-budget 0 > $O/bench_pair${v}_1s.json 2> $O/bench_pair${v}_1s.err
done
tail -3 $O/pytest.log; grep resid $O/gemm_shapes_pair*.log; for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['parity_max_abs_err_vs_oracle'])" 2>&1 | tail -1; done
