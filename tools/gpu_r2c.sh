#!/bin/bash
# round-2 GPU session C: exact-ranking batched dense top-k (query hi/lo split) + top-k probe
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2c; mkdir -p $O
timeout 900 python -m pytest tests/test_topk_gpu.py tests/test_query_batch_gpu.py tests/test_sharded_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 600 python tools/bench_topk.py > $O/topk_lines.json 2> $O/topk.err
tail -15 $O/pytest.log; cat $O/topk_lines.json | cut -c1-260
