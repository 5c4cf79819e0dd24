#!/bin/bash
# round-2 GPU session U: banded attention with compile-time tile patterns -- parity, bench
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2u; mkdir -p $O
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_fuzz_gpu.py tests/test_full_shapes_gpu.py tests/test_heads_gpu.py tests/test_extractor_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
timeout 300 python bench.py --cpu-budget 0 --steps 10 > $O/bench.json 2>/dev/null
python - <<'PY' | tee gpurun_out/r2u/summary.txt
import json
d=json.loads(open('gpurun_out/r2u/bench.json').read().strip().splitlines()[-1])
print(round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['isolated_pass']['breakdown_ms_per_step'])
PY
