#!/usr/bin/env python3
"""Kernel-tuning helper: TFLOP/s of the GEMM instantiations on the ModernBERT-base shapes."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verbatim_rag_amd  # noqa
from verbatim_rag_amd import _lib

EPI = {"f32": 0, "bf16": 1, "gelu": 2, "resid": 3, "geglu": 4, "qkv": 5, "none": 7}
lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
for name, epi, N, K in [("none  N2304 K768", 7, 2304, 768), ("qkv   N2304 K768", 5, 2304, 768),
                        ("geglu N2304 K768", 4, 2304, 768), ("none  N768  K768", 7, 768, 768),
                        ("resid N768  K768", 3, 768, 768), ("none  N768 K1152", 7, 768, 1152),
                        ("resid N768 K1152", 3, 768, 1152), ("bf16  N2304 K768", 1, 2304, 768)]:
    ms = C.c_float()
    _lib.check("gemm", lib.vrag_debug_gemm_ms(epi, M, N, K, 20, 0, C.byref(ms)))
    print(f"{name}  M={M}: {ms.value*1e3:8.1f} us  {2.0*M*N*K/ms.value/1e9:8.1f} TFLOP/s")
