#!/bin/bash
# round-2 GPU session AH: distance between the two dependent MFMAs of an accumulator: compiler's choice (4) vs adjacent (two fence masks)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2ah; mkdir -p $O
for rep in 1 2; do
for lib in "" adj0 adj0x100; do
  echo "== ${lib:-product}"
  if [ -n "$lib" ]; then export VRAG_AMD_LIB=$PWD/verbatim-rag_amd/build/libvrag_$lib.so; else unset VRAG_AMD_LIB; fi
  timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep "none"
done; done | tee $O/adj.txt
