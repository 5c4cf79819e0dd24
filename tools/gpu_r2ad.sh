#!/bin/bash
# round-2 GPU session AD: final tree -- full GPU suite, smoke, default bench line
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2ad; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err
tail -3 $O/pytest.log; tail -1 $O/smoke.log; tail -c 600 $O/bench_line.json
