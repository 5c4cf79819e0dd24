#!/usr/bin/env python3
"""fp32-row dense shard: per-query cost of the 4-queries-per-pass LDS path vs single-query passes."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import verbatim_rag_amd  # noqa: F401
from verbatim_rag_amd.vector_stores import DenseShard

n, dim, k = 500_000, 768, 10
rng = np.random.default_rng(0)
sh = DenseShard(dim, n, "f32")
for _ in range(n // 125_000):
    sh.add((rng.integers(-64, 65, size=(125_000, dim)) / 64.0).astype(np.float32))
for nq in (1, 2, 4, 8):
    q = (rng.integers(-64, 65, size=(nq, dim)) / 64.0).astype(np.float32)
    sh.search(q, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        sh.run_resident(nq, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(json.dumps({"nq": nq, "ms": dt * 1e3, "ms_per_query": dt * 1e3 / nq, "GBps_per_pass_equiv": n * dim * 4 * nq / dt / 1e9}))
sh.close()
