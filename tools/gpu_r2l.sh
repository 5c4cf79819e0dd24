#!/bin/bash
# round-2 GPU session L: micro-batch / stream sweep with the round-2 kernels; ModernBERT-large bench line
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2l; mkdir -p $O
for mb in 32768 65536 131072; do for st in 2 3; do
  VRAG_STREAMS=$st timeout 300 python bench.py --cpu-budget 0 --no-profile --steps 8 --micro-batch-tokens $mb > $O/b_${mb}_$st.json 2>/dev/null
  python -c "import json; d=json.loads(open('$O/b_${mb}_$st.json').read().strip().splitlines()[-1]); print('mb', $mb, 'streams', $st, round(d['value']), round(d['ms_per_step'],2))"
done; done
timeout 600 python bench.py --cpu-budget 0 --model large > $O/bench_large.json 2> $O/bench_large.err
python -c "import json; d=json.loads(open('$O/bench_large.json').read().strip().splitlines()[-1]); print('large', round(d['value']), round(d['ms_per_step'],2), d['model_mfma_frac'])"
