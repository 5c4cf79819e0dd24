#!/usr/bin/env python3
"""Probe (round 5): the tiled dense search's GEMM tile per query count.  Needs a build whose dense_tiled_search reads
VRAG_TOPK_TILE (the shipped one chooses: 256 x 128 tiles on a three-stage ring up to 128 query columns, 256 x 256 x 2 above).
Recorded: profiles/r05_tiled_topk_tile_probe.txt."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import verbatim_rag_amd  # noqa: E402,F401
from verbatim_rag_amd import _lib  # noqa: E402
from verbatim_rag_amd.vector_stores import DenseShard  # noqa: E402

lib = _lib.load()
n, dim, k = 1_250_000, 768, 10
rng = np.random.default_rng(0)
sh = DenseShard(dim, n, "bf16")
for _ in range(n // 125_000):
    sh.add((rng.integers(-64, 65, size=(125_000, dim)) / 64.0).astype(np.float32))
for nq in (64, 128, 256, 1024, 4096):
    q = (rng.integers(-64, 65, size=(nq, dim)) / 64.0).astype(np.float32)
    ref = None
    for name, tile in (("256x256x2", "0"), ("256x128x3", "1")):
        os.environ["VRAG_TOPK_TILE"] = tile
        s, i = sh.search(q, k)
        if ref is None:
            ref = (s, i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            sh.run_resident(nq, k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(json.dumps({"nq": nq, "gemm_config": name, "ms": dt * 1e3, "GBps": n * dim * 2 / dt / 1e9,
                          "same_result": bool(np.array_equal(s, ref[0]) and np.array_equal(i, ref[1]))}), flush=True)
sh.close()
