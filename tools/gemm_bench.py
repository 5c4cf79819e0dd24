#!/usr/bin/env python3
"""Kernel-tuning helper: TFLOP/s of the GEMM instantiations (HIP events around `iters` back-to-back launches, synthetic
pseudo-random [-1, 1) bf16 operands).  No torch import.
  python tools/gemm_bench.py                      ModernBERT-base shapes at M = 131072, 20 launches each
  python tools/gemm_bench.py 65536                same at M = 65536 (one micro-batch)
  python tools/gemm_bench.py cal [iters]          calibration against the guide's 256^2 template: EPI_NONE at 4096^3 / 8192^3
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verbatim_rag_amd  # noqa
from verbatim_rag_amd import _lib

lib = _lib.load()

dbg = _lib.load_debug()   # harness library (include/vrag_amd_debug.h)


def run(name, epi, M, N, K, iters):
    ms = C.c_float()
    _lib.check_debug("gemm", dbg.vrag_debug_gemm_ms(epi, M, N, K, iters, 0, C.byref(ms)))
    print(f"{name}  M={M} N={N} K={K} x{iters}: {ms.value*1e3:8.1f} us  {2.0*M*N*K/ms.value/1e9:8.1f} TFLOP/s", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "cal":
    dbg.vrag_set_small_batch_rows(0)   # M = 4096 / 8192 would otherwise take the small-batch configuration
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    for n in (4096, 8192):
        run("none ", 7, n, n, n, iters)
        run("bf16 ", 1, n, n, n, iters)
else:
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    for name, epi, N, K in [("none ", 7, 2304, 768), ("qkv  ", 5, 2304, 768), ("geglu", 4, 2304, 768), ("none ", 7, 768, 768),
                            ("resid", 3, 768, 768), ("none ", 7, 768, 1152), ("resid", 3, 768, 1152), ("bf16 ", 1, 2304, 768)]:
        run(name, epi, M, N, K, 50)
