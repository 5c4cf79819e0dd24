#!/bin/bash
# round-2 GPU session AE: MFMA order probe -- the two k-substeps of an accumulator adjacent (dependent pairs) vs 31 MFMAs apart (EPI_NONE)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2ae; mkdir -p $O
for rep in 1 2; do
echo "== product order"; timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep "none"; timeout 300 python tools/gemm_bench.py cal 50 2>&1 | grep none
echo "== chained"; VRAG_AMD_LIB=$PWD/verbatim-rag_amd/build/libvrag_chain.so timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep "none"; VRAG_AMD_LIB=$PWD/verbatim-rag_amd/build/libvrag_chain.so timeout 300 python tools/gemm_bench.py cal 50 2>&1 | grep none
done | tee $O/chain.txt
