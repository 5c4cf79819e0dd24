#!/bin/bash
# round-2 GPU session E: new full-shape tests, attention output staging, f16 bench line, round profile (kernel trace + PMC)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests/test_full_shapes_gpu.py tests/test_encoder_gpu.py tests/test_extractor_gpu.py tests/test_bert_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 600 python bench.py --cpu-budget 0 --operand-dtype f16 > $O/bench_f16.json 2> $O/bench_f16.err
timeout 1500 bash tools/profile_round.sh r02_v1 > $O/profile.log 2>&1
tail -5 $O/pytest.log; python -c "import json; d=json.loads(open('$O/bench_f16.json').read().strip().splitlines()[-1]); print('f16', d['value'], d['ms_per_step'], d['parity_max_abs_err_vs_oracle'])"; tail -30 $O/profile.log
