"""fp32 rows (1.25 M x 768, bf16 prefilter image), public call, k = 10: ms per search for 1-8 queries (2-4: one image pass for the batch)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import verbatim_rag_amd
from verbatim_rag_amd.vector_stores import DenseShard
n, dim, k = 1_250_000, 768, 10
rng = np.random.default_rng(0)
sh = DenseShard(dim, n, "f32")
for _ in range(n // 125_000):
    sh.add(rng.standard_normal((125_000, dim)).astype(np.float32))
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for nq in (1, 2, 3, 4, 5, 8, 16, 32):
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    dt = timed(lambda: sh.search(q, k))
    d_s = torch.empty((nq, k), dtype=torch.float32, device="cuda"); d_i = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    dd = timed(lambda: sh.search_device(q, k, d_s.data_ptr(), d_i.data_ptr(), stream=None))
    print(json.dumps({"nq": nq, "search_ms": round(dt * 1e3, 3), "search_device_ms": round(dd * 1e3, 3)}))
sh.close()
