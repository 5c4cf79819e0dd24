#!/bin/bash
# round-2 GPU session AA: soak (handle lifetimes, device-memory drift) and a long steady bench with the final build
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2aa; mkdir -p $O
timeout 600 python tools/soak.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/soak.txt
timeout 600 python bench.py --cpu-budget 0 --no-profile --steps 600 --warmup 20 2>/dev/null | tail -1 | cut -c1-330 | tee $O/bench_600.json
