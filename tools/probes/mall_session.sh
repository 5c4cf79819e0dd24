#!/bin/bash
# MALL probe session: the residual GEMM's epilogue with and without the non-temporal hint, at a working set below and above the
# 256 MB Infinity Cache, with package power.  The plain build: VRAG_BUILD_VARIANT=plain VRAG_HIPCC_FLAGS=-DVRAG_PLAIN_STREAMS python verbatim-rag_amd/build.py
for lib in "" "$PWD/verbatim-rag_amd/libvrag_amd_plain.so"; do
  for tok in 21760 65280; do
    echo "## lib=${lib:-default} tokens=$tok"
    VRAG_AMD_LIB=${lib:-$PWD/verbatim-rag_amd/libvrag_amd.so} python tools/energy_by_class.py --seconds 2.5 --tokens $tok \
      --only 'resid epilogue-only,resid plain,gemm_wo (,geglu epilogue' --out gpurun_out/r4d/e_$(basename ${lib:-default} .so)_$tok.json 2>&1 | grep -v "amdgpu.ids\|^{"
  done
  echo "## lib=${lib:-default} mall_probe"
  VRAG_AMD_LIB=${lib:-$PWD/verbatim-rag_amd/libvrag_amd.so} python tools/probes/mall_probe.py 2>&1 | grep -v amdgpu.ids | grep "resid\|geglu"
done
