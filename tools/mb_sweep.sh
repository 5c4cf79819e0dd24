#!/bin/bash
# micro-batch sweep of bench.py (tuning helper): prints value, ms/step and the single-stream breakdown
for mb in "$@"; do
  python bench.py --cpu-budget 0 --micro-batch-tokens $mb 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
b = d['roofline']['breakdown_ms_per_step']
print($mb, round(d['value'], 1), round(d['ms_per_step'], 2), {k: round(v, 2) for k, v in b.items()})"
done
