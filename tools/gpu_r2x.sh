#!/bin/bash
# round-2 GPU session X: batched sparse top-k with 16-term steps -- parity, rates with QB = 16 allowed / forced to 8
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2x; mkdir -p $O
timeout 900 python -m pytest tests/test_topk_gpu.py tests/test_query_batch_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_gpu.txt
for qb8 in "" 1; do
  echo "== VRAG_SPARSE_QB8=$qb8"
  if [ -n "$qb8" ]; then export VRAG_SPARSE_QB8=1; fi
  timeout 600 python tools/bench_topk.py 2>/dev/null | grep '^{' | grep '"sparse' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(' nq', d['nq'], 'ms', round(d['ms'],3), 'GB/s per pass', round(d['algorithmic_GBps']), 'passes', d['passes'], 'q/s', round(d['queries_per_s']))
"
done | tee $O/sparse.txt
