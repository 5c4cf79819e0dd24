#!/bin/bash
# round-2 GPU session H: bit-exact fp32-row top-k with the queries in registers -- parity + speed
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2h; mkdir -p $O
timeout 900 python -m pytest tests/test_topk_gpu.py tests/test_query_batch_gpu.py tests/test_sharded_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 600 python tools/bench_dense_f32.py > $O/dense_f32_exact2.log 2>&1
VRAG_TOPK_EXACT_LDSQ=1 timeout 600 python tools/bench_dense_f32.py > $O/dense_f32_exact_ldsq.log 2>&1
tail -6 $O/pytest.log; cat $O/dense_f32_exact2.log | cut -c1-330 | tail -6; cat $O/dense_f32_exact_ldsq.log | cut -c1-330 | tail -6
