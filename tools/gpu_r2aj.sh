#!/bin/bash
# round-2 GPU session AJ: MFMA loop order probe -- row tile outermost (product: consecutive MFMAs share the activation fragment) vs
# column tile outermost (they share the weight fragment); EPI_NONE
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2aj; mkdir -p $O
for rep in 1 2; do
for lib in "" nj; do
  echo "== ${lib:-product}"
  if [ -n "$lib" ]; then export VRAG_AMD_LIB=$PWD/verbatim-rag_amd/build/libvrag_$lib.so; else unset VRAG_AMD_LIB; fi
  timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep "none"
done; done | tee $O/nj.txt
