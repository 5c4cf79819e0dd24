#!/bin/bash
# round-2 GPU session F: 8-phase GEMM schedule A/B (VRAG_GEMM_SCHED8) -- speed on the encoder shapes / calibration shapes, parity
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2f; mkdir -p $O
for v in 0 1; do
  VRAG_GEMM_SCHED8=$v timeout 300 python tools/gemm_bench.py 65536 > $O/gemm_shapes_s$v.log 2>&1
  VRAG_GEMM_SCHED8=$v timeout 300 python tools/gemm_bench.py cal 100 > $O/gemm_cal_s$v.log 2>&1
done
VRAG_GEMM_SCHED8=1 timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_full_shapes_gpu.py tests/test_extractor_gpu.py -m gpu -q -x > $O/pytest_s1.log 2>&1; echo "rc=$?" >> $O/pytest_s1.log
VRAG_GEMM_SCHED8=1 timeout 600 python bench.py --cpu-budget 0 > $O/bench_s1.json 2> $O/bench_s1.err
VRAG_GEMM_SCHED8=0 timeout 600 python bench.py --cpu-budget 0 > $O/bench_s0.json 2> $O/bench_s0.err
paste $O/gemm_shapes_s0.log $O/gemm_shapes_s1.log | cut -c1-220; paste $O/gemm_cal_s0.log $O/gemm_cal_s1.log | cut -c1-220; tail -5 $O/pytest_s1.log
for f in $O/bench_s*.json; do python -c "import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
