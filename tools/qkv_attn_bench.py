"""Phase decomposition of the fused Wqkv + RoPE + attention kernel (csrc/qkv_attn.hip) on one 65 536-token micro-batch of
512-token sequences: whole kernel, without the attention phase, without the main-loop MFMAs, DMA + barriers only, and the
two-kernel path it replaces (QKV GEMM + attention launch) from the same process.  Short bursts (not power-limited)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from verbatim_rag_amd import _lib  # noqa: E402

lib = _lib.load()

dbg = _lib.load_debug()   # harness library (include/vrag_amd_debug.h)
n_seqs, S, H = int(os.environ.get("NSEQ", 128)), int(os.environ.get("S", 512)), int(os.environ.get("H", 768))
iters = int(os.environ.get("ITERS", 50))


def fused(local, flags):
    ms = C.c_float()
    rc = dbg.vrag_debug_qkv_attn_ms(local, n_seqs, S, H, 64, iters, flags, 0, C.byref(ms))
    assert rc == 0, lib.vrag_last_error()
    return ms.value * 1e3


def gemm(epi, M, N, K):
    ms = C.c_float()
    assert dbg.vrag_debug_gemm_ms(epi, M, N, K, iters, 0, C.byref(ms)) == 0
    return ms.value * 1e3


def attn(local):
    ms = C.c_float()
    assert dbg.vrag_debug_attn_ms(local, n_seqs, S, H, 64, iters, 0, C.byref(ms)) == 0
    return ms.value * 1e3


for local in (0, 1):
    name = "banded" if local else "global"
    full, noattn, nomfma, dma_only, nothing = fused(local, 0), fused(local, 1), fused(local, 2), fused(local, 3), fused(local, 7)
    print(f"{name}: fused {full:7.1f} us | no attention {noattn:7.1f} | no main-loop MFMA {nomfma:7.1f} | DMA+barriers+epilogue {dma_only:7.1f} | "
          f"epilogue only {nothing:7.1f}", flush=True)
    print(f"{name}: MFMA loop without DMA + epilogue {fused(local, 5):7.1f} | the same + attention {fused(local, 4):7.1f} | barriers + epilogue + attention {fused(local, 6):7.1f}", flush=True)
    print(f"{name}: launch + empty main loop {fused(local, 7 + 16):7.1f} | epilogue without the O store {fused(local, 7 + 8):7.1f}", flush=True)
    two = gemm(5, n_seqs * S, 3 * H, H), attn(local)
    print(f"{name}: two kernels: QKV GEMM {two[0]:7.1f} us + attention {two[1]:7.1f} us = {sum(two):7.1f} us", flush=True)
