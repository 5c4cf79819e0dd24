#!/bin/bash
# round-2 GPU session V: batched sparse top-k ([union id][q] weights, prefetched term batches) -- parity, rates
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2v; mkdir -p $O
timeout 900 python -m pytest tests/test_topk_gpu.py tests/test_query_batch_gpu.py tests/test_sharded_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
timeout 600 python tools/bench_topk.py 2>/dev/null | grep '^{' | grep sparse | cut -c1-330 | tee $O/topk_sparse.json
