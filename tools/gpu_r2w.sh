#!/bin/bash
# round-2 GPU session W: one 131072-token micro-batch on one stream vs two 65536-token micro-batches on two streams, final kernels
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2w; mkdir -p $O
for rep in 1 2 3; do
  for cfg in "65536 2" "131072 1" "65536 1" "262144 1"; do
    set -- $cfg
    VRAG_STREAMS=$2 timeout 300 python bench.py --cpu-budget 0 --no-profile --steps 12 --micro-batch-tokens $1 2>/dev/null | tail -1 > $O/b.json
    python -c "import json; d=json.loads(open('$O/b.json').read()); print('rep', $rep, 'mb', $1, 'streams', $2, round(d['value'],1), round(d['ms_per_step'],3))"
  done
done | tee $O/sweep.txt
