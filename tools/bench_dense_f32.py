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
def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for nq in (1, 2, 4, 32, 256):
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    sh.search(q, k)
    dt = timed(lambda: sh.run_resident(nq, k))           # the full fp32 scan on resident queries (no copies, no sync)
    per_pass = 32 if exact else (1 if nq == 1 else 4)
    passes = (nq + per_pass - 1) // per_pass
    print(json.dumps({"kind": "dense_f32" + ("_exact_mfma" if exact else "_scalar"), "rows": n, "dim": dim, "nq": nq, "k": k,
                      "ms": dt * 1e3, "queries_per_s": nq / dt, "passes": passes, "bytes_per_pass": n * dim * 4,
                      "algorithmic_GBps": n * dim * 4 * passes / dt / 1e9, "frac_of_8TBps": n * dim * 4 * passes / dt / 8e12}))
    # the public call (query upload, kernels, read-back): with the prefilter image (the default) 1-2 and >= 64 queries rank the
    # bf16 image for 64 candidates and re-score them exactly; without it the same call runs the scan above
    dt_api = timed(lambda: sh.search(q, k))
    print(json.dumps({"kind": "dense_f32_search_call", "prefilter_image": True, "rows": n, "dim": dim, "nq": nq, "k": k,
                      "ms": dt_api * 1e3, "queries_per_s": nq / dt_api,
                      "route": "bf16 image, one pass for the batch: prefix thresholds -> candidates -> exact re-score" if nq <= 4 else ("bf16 image, tiled search in collect form: prefix thresholds -> one appending pass -> exact re-score" if nq <= 256 else "bf16 image, tiled search -> 64 candidates -> exact re-score"),   # round 6: one to four queries share a pass (dim % 256 == 0), larger batches are tiled (csrc/topk.hip pf_onepass_max)
                      "image_bytes": n * dim * 2}))
# the device-resident search a rank of the sharded store runs per batch (lists left in HBM, no host round trip): prefilter route with
# the full scan behind the per-query flags
for nq in (1, 2, 256, 1024):
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    d_s = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    d_i = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    dt_dev = timed(lambda: sh.search_device(q, k, d_s.data_ptr(), d_i.data_ptr()))
    print(json.dumps({"kind": "dense_f32_search_device", "prefilter_image": True, "rows": n, "dim": dim, "nq": nq, "k": k,
                      "ms": dt_dev * 1e3, "queries_per_s": nq / dt_dev,
                      "full_scan_would_be_ms": 0.97 * ((nq + 31) // 32)}))
sh.close()
