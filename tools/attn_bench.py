#!/usr/bin/env python3
"""Attention kernels alone (vrag_debug_attn_ms): us per launch at ModernBERT-base shapes, one 65 536-token micro-batch.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verbatim_rag_amd  # noqa
from verbatim_rag_amd import _lib

lib = _lib.load()

dbg = _lib.load_debug()   # harness library (include/vrag_amd_debug.h)
tag = "attention"
for name, local, n_seqs, S in [("global S=512", 0, 128, 512), ("banded S=512", 1, 128, 512), ("global S=8192", 0, 8, 8192),
                               ("banded S=8192", 1, 8, 8192), ("global S=200", 0, 320, 200), ("banded S=200", 1, 320, 200)]:
    ms = C.c_float()
    _lib.check_debug("attn", dbg.vrag_debug_attn_ms(local, n_seqs, S, 768, 64, 200, 0, C.byref(ms)))
    flop = 4.0 * n_seqs * S * (min(S, 129) if local else S) * 768
    print(f"{tag} {name:16s} {ms.value * 1e3:8.1f} us  {flop / (ms.value * 1e-3) / 1e12:7.1f} TFLOP/s", flush=True)
