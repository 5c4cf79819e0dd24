#!/bin/bash
# round-2 GPU session Z: final evidence of the round -- full suite, smoke, bench lines (bf16 / f16 / 2-rank), kernel trace + PMC,
# aux probes (top-k, embedding, fp32 rows), GEMM rates, end-to-end compositions, energy line
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2z; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 1800 bash tools/profile_round.sh r02 > $O/profile.log 2>&1
timeout 600 python bench.py --cpu-budget 0 --operand-dtype f16 > $O/bench_f16.json 2> $O/bench_f16.err
VRAG_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --cpu-budget 0 --steps 3 > $O/bench_g2.json 2> $O/bench_g2.err
timeout 600 python bench.py --cpu-budget 0 --model large > $O/bench_large.json 2> $O/bench_large.err
timeout 1500 bash tools/profile_aux.sh r02 > $O/profile_aux.log 2>&1
timeout 300 python tools/gemm_bench.py 65536 > $O/gemm_shapes.log 2>&1
timeout 300 python tools/gemm_bench.py cal 100 > $O/gemm_cal.log 2>&1
timeout 900 python tools/bench_e2e.py > $O/e2e.json 2> $O/e2e.err
timeout 900 python tools/bench_e2e.py --hybrid > $O/e2e_hybrid.json 2> $O/e2e_hybrid.err
timeout 900 python tools/bench_api.py > $O/api.json 2> $O/api.err
timeout 600 python tools/energy_probe.py --steps 300 2>&1 | tail -1 > $O/energy.json
tail -4 $O/pytest.log; cat $O/smoke.log | tail -2; tail -3 $O/e2e.json | cut -c1-300
