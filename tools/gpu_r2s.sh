#!/bin/bash
# round-2 GPU session S: operand-DMA issue split (VRAG_DMA_SPLIT 1 vs 0) with two k-substeps per stage
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r2s; mkdir -p $O
for rep in 1 2; do
echo "== split 1 (in-tree library)"; timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep -v amdgpu.ids
echo "== split 0"; VRAG_AMD_LIB=$PWD/verbatim-rag_amd/build/libvrag_split0.so timeout 300 python tools/gemm_bench.py 65536 2>&1 | grep -v amdgpu.ids
done | tee $O/split.txt
