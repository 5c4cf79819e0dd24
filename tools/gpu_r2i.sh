#!/bin/bash
# round-2 GPU session I: pipelined exact fp32 top-k, skewed banded attention; full suite + bench
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2i; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 600 python tools/bench_dense_f32.py > $O/dense_f32_exact2.log 2>&1
timeout 600 python bench.py --cpu-budget 0 > $O/bench.json 2> $O/bench.err
tail -6 $O/pytest.log; cat $O/dense_f32_exact2.log | cut -c1-330 | tail -6
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print({k:round(v,2) for k,v in d['roofline']['isolated_pass']['breakdown_ms_per_step'].items()})"
