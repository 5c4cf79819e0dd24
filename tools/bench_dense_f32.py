#!/usr/bin/env python3
"""fp32-row dense shard (the store's default dtype): ms per search and algorithmic GB/s per pass for 1..256 resident
queries.  Default = the bit-exact fp32-MFMA kernel (32 queries per pass); VRAG_TOPK_NO_EXACT=1 = the scalar kernels
(1 query in registers / 4 in LDS per pass)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (device sync only)

import verbatim_rag_amd  # noqa: E402,F401
from verbatim_rag_amd.vector_stores import DenseShard  # noqa: E402

n, dim, k = 1_250_000, 768, 10
exact = not os.environ.get("VRAG_TOPK_NO_EXACT")
rng = np.random.default_rng(0)
sh = DenseShard(dim, n, "f32")
for _ in range(n // 125_000):
    sh.add(rng.standard_normal((125_000, dim)).astype(np.float32))
for nq in (1, 2, 4, 32, 256):
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    sh.search(q, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        sh.run_resident(nq, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    per_pass = 32 if exact else (1 if nq == 1 else 4)
    passes = (nq + per_pass - 1) // per_pass
    print(json.dumps({"kind": "dense_f32" + ("_exact_mfma" if exact else "_scalar"), "rows": n, "dim": dim, "nq": nq, "k": k,
                      "ms": dt * 1e3, "queries_per_s": nq / dt, "passes": passes, "bytes_per_pass": n * dim * 4,
                      "algorithmic_GBps": n * dim * 4 * passes / dt / 1e9, "frac_of_8TBps": n * dim * 4 * passes / dt / 8e12}))
sh.close()
