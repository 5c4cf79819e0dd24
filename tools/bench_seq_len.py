"""Encoder throughput by sequence length (131 072 tokens per step, uniform lengths): which path serves short chunks better --
the fused QKV + attention kernel (one 512-token workgroup per sequence and head, idle waves past the end of a short sequence)
or the two-kernel path over packed rows?  Run once per setting of VRAG_FUSED_QKV_ATTN; prints tokens/s per length."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape  # noqa: E402
from verbatim_rag_amd.weights import random_init  # noqa: E402

shape = ModernBertShape.base()
w = random_init(shape, seed=1234)
rng = np.random.default_rng(0)
out = {"fused": os.environ.get("VRAG_FUSED_QKV_ATTN", "1 (default)")}
for L in [int(x) for x in os.environ.get("LENS", "64,128,192,256,320,384,448,512").split(",")]:
    n = 131072 // L
    seqs = [rng.integers(1000, 50000, size=L).astype(np.int32) for _ in range(n)]
    eng = EncoderEngine(shape, w, max_tokens=n * L, max_seqs=n, max_seq_len=512, max_ranges=16, micro_batch_tokens=65536)
    eng.load_batch(seqs)
    for _ in range(2):
        eng.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    out[str(L)] = {"ms": round(dt * 1e3, 2), "ktok_per_s": round(n * L / dt / 1e3, 1)}
    eng.close()
print(json.dumps(out))
