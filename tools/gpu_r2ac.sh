#!/bin/bash
# round-2 GPU session AC: A/B of the split residual stream against the previous commit's library in one session
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2ac; mkdir -p $O
for rep in 1 2 3; do
  for lib in new prev; do
    if [ $lib = prev ]; then export VRAG_AMD_LIB=$PWD/verbatim-rag_amd/build/libvrag_prev.so; else unset VRAG_AMD_LIB; fi
    timeout 300 python bench.py --cpu-budget 0 --no-profile --steps 12 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), round(d['ms_per_step'],3))"
  done
done | tee $O/ab.txt
unset VRAG_AMD_LIB
timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep "resid" | sed 's/^/new  /' | tee -a $O/ab.txt
VRAG_AMD_LIB=$PWD/verbatim-rag_amd/build/libvrag_prev.so timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep "resid" | sed 's/^/prev /' | tee -a $O/ab.txt
