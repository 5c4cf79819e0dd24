#!/bin/bash
# round-2 GPU session AG: would 32x32x16 MFMAs, chained four per accumulator tile, beat the chained 16x16x32 loop?  (EPI_NONE probe, garbage results)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2ag; mkdir -p $O
for rep in 1 2; do
echo "== product (16x16x32, pairs chained)"; timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep "none"; timeout 300 python tools/gemm_bench.py cal 50 2>&1 | grep none
echo "== probe (32x32x16, four chained)"; VRAG_AMD_LIB=$PWD/verbatim-rag_amd/build/libvrag_32c.so timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep "none"; VRAG_AMD_LIB=$PWD/verbatim-rag_amd/build/libvrag_32c.so timeout 300 python tools/gemm_bench.py cal 50 2>&1 | grep none
done | tee $O/probe.txt
