#!/usr/bin/env python3
"""Attention kernels alone against a float64 softmax on the same bf16 operands (vrag_debug_attn_run): worst error per
sequence length and mode, and -- for the banded mode -- the mean error by (row mod 32), which is where a masking or
fragment-mapping slip shows."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verbatim_rag_amd  # noqa
from verbatim_rag_amd import _lib

lib = _lib.load()

dbg = _lib.load_debug()   # harness library (include/vrag_amd_debug.h)


def bf16_bits(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) >> 16).astype(np.uint16)


def from_bits(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def run(local, n_seqs, S, H=128, W=64, seed=0, sharp=1.0):
    rng = np.random.default_rng(seed)
    T, nh = n_seqs * S, H // 64
    Tp = (T + 255) // 256 * 256
    q = bf16_bits(rng.standard_normal((T, H)) * sharp * 0.125 * 1.4426950408889634)
    k = bf16_bits(rng.standard_normal((T, H)))
    v = bf16_bits(rng.standard_normal((T, H)))
    vt = np.zeros((H, Tp), np.uint16)
    vt[:, :T] = v.T
    o = np.zeros((T, H), np.uint16)
    _lib.check_debug("attn", dbg.vrag_debug_attn_run(local, n_seqs, S, H, W, 0, q.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p),
                                              vt.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), 0))
    got = from_bits(o).astype(np.float64)
    qf, kf, vf = from_bits(q).astype(np.float64), from_bits(k).astype(np.float64), from_bits(v).astype(np.float64)
    err = np.zeros(T)
    for s in range(n_seqs):
        for h in range(nh):
            sl, hs = slice(s * S, (s + 1) * S), slice(h * 64, (h + 1) * 64)
            sc = qf[sl, hs] @ kf[sl, hs].T                      # log2 units (q carries the scale)
            if local:
                i = np.arange(S)
                sc = np.where(np.abs(i[:, None] - i[None, :]) <= W, sc, -np.inf)
            p = np.exp2(sc - sc.max(1, keepdims=True))
            ref = (p / p.sum(1, keepdims=True)) @ vf[sl, hs]
            err[sl] = np.maximum(err[sl], np.abs(got[sl, hs] - ref).max(1))
    return err


if __name__ == "__main__":
    tag = "attention"
    for local in (0, 1):
        for S in (64, 200, 512, 1000):
            for sharp in (1.0, 6.0):
                err = run(local, 2, S, seed=S + local, sharp=sharp)
                line = f"{tag} {'banded' if local else 'global'} S={S:4d} sharp={sharp}: max {err.max():.2e} mean {err.mean():.2e}"
                if local:
                    per = np.asarray([err[i::32].mean() for i in range(32)])
                    line += "  mean by row%32: " + " ".join(f"{x * 1e3:.1f}" for x in per) + " (x1e-3)"
                print(line, flush=True)
