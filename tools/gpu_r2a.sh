#!/bin/bash
# round-2 GPU session A: parity suite, GEMM calibration, residual-tile / stream-count A/B, 2-rank bench self-test
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 300 python tools/gemm_bench.py cal 200 > $O/gemm_cal.log 2>&1
timeout 300 python tools/gemm_bench.py 65536 > $O/gemm_shapes.log 2>&1
VRAG_GEMM_RES_TILE128=1 timeout 300 python tools/gemm_bench.py 65536 > $O/gemm_shapes_res128.log 2>&1
timeout 600 python bench.py --cpu-budget 0 > $O/bench_2s.json 2> $O/bench_2s.err
VRAG_STREAMS=1 timeout 600 python bench.py --cpu-budget 0 > $O/bench_1s.json 2> $O/bench_1s.err
VRAG_GEMM_RES_TILE128=1 timeout 600 python bench.py --cpu-budget 0 > $O/bench_res128.json 2> $O/bench_res128.err
VRAG_GEMM_RES_TILE128=1 VRAG_STREAMS=1 timeout 600 python bench.py --cpu-budget 0 > $O/bench_res128_1s.json 2> $O/bench_res128_1s.err
VRAG_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --cpu-budget 0 --steps 3 > $O/bench_g2.json 2> $O/bench_g2.err
tail -3 $O/pytest.log; cat $O/gemm_cal.log; for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['n_gpus'], d.get('sharded_topk'))" 2>&1 | tail -1; done
