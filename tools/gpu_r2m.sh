#!/bin/bash
# round-2 GPU session M: vendor GEMM reference point (rates + which tiling it picks), round-2 energy line
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2m; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/vendor_gemm_probe.py 50 2>&1 | tee $O/vendor.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/vprof -- python tools/vendor_gemm_probe.py 5 > /dev/null 2>$O/vprof.err
python tools/rocpd_summary.py "$(find $O/vprof -name "*.db" | head -1)" 2>&1 | cut -c1-400 | head -30 | tee $O/vendor_kernels.txt
timeout 600 python tools/energy_probe.py --steps 300 2>&1 | tail -3 | tee $O/energy.txt
timeout 300 python tools/gemm_bench.py 65536 2>&1 | tee $O/gemm_bench.txt
