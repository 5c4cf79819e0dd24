#!/usr/bin/env python3
"""Does the 256 MB Infinity Cache hold a micro-batch's activations between launches?  The residual GEMM looped alone at row
counts whose working set (fp32 stream + 16-bit copy + A operand) is below / around / above 256 MB; one full tile round = 85 row
blocks of 256 rows (255 tiles of 256 x 256 on 256 CUs).  Prints us per launch and us per tile round."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import verbatim_rag_amd  # noqa: E402,F401
from verbatim_rag_amd import _lib  # noqa: E402

lib = _lib.load()

dbg = _lib.load_debug()   # harness library (include/vrag_amd_debug.h)
H, I = 768, 1152
for epi, name, N, K in ((3, "resid", H, 64), (3, "resid", H, H), (3, "resid", H, I), (4, "geglu", 2 * I, H), (7, "none", H, H)):
    for rounds in (1, 2, 3, 6):
        M = 85 * 256 * rounds
        ms = C.c_float()
        _lib.check_debug("gemm", dbg.vrag_debug_gemm_ms(epi, M, N, K, 300, 0, C.byref(ms)))
        ws = M * (H * 4 + H * 2 + K * 2) / 1e6 if epi == 3 else M * (K * 2 + N) / 1e6
        print(f"{name:6s} N={N:5d} K={K:5d} M={M:6d} ({rounds} rounds, working set {ws:6.0f} MB): {ms.value * 1e3:7.1f} us = "
              f"{ms.value * 1e3 / rounds:6.1f} us / round", flush=True)
