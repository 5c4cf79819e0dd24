#!/bin/bash
# round-2 GPU session K: banded attention per-half skip; embedding / end-to-end probes with the round-2 build
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2k; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 600 python bench.py --cpu-budget 0 > $O/bench.json 2> $O/bench.err
timeout 600 python tools/bench_embed.py > $O/embed_lines.json 2> $O/embed.err
timeout 900 python tools/bench_e2e.py > $O/e2e.json 2> $O/e2e.err
timeout 900 python tools/bench_e2e.py --hybrid > $O/e2e_hybrid.json 2> $O/e2e_hybrid.err
timeout 900 python tools/bench_api.py > $O/api.json 2> $O/api.err
tail -4 $O/pytest.log
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print({k:round(v,2) for k,v in d['roofline']['isolated_pass']['breakdown_ms_per_step'].items()})"
cat $O/embed_lines.json | cut -c1-300; tail -2 $O/e2e.json | cut -c1-400; tail -2 $O/e2e_hybrid.json | cut -c1-400; tail -2 $O/api.json | cut -c1-400; tail -3 $O/e2e.err $O/api.err
