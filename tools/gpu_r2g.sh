#!/bin/bash
# round-2 GPU session G: bit-exact fp32-row dense top-k (fp32 MFMA) -- parity + speed vs the scalar kernels
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2g; mkdir -p $O
timeout 900 python -m pytest tests/test_topk_gpu.py tests/test_query_batch_gpu.py tests/test_sharded_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 600 python tools/bench_dense_f32.py > $O/dense_f32_exact.log 2>&1
VRAG_TOPK_NO_EXACT=1 timeout 600 python tools/bench_dense_f32.py > $O/dense_f32_scalar.log 2>&1
tail -25 $O/pytest.log; cat $O/dense_f32_exact.log | tail -8; cat $O/dense_f32_scalar.log | tail -8
