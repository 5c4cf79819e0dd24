"""A/B of the tiled dense search's tile for <= 64 query columns (VRAG_TOPK_TILE64=0: 256 x 128 x 3 stages, default: 256 x 64 x 4 stages).
bf16 rows, device-resident queries; exact bf16 queries (one column each) and generic fp32 queries (column pairs)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import verbatim_rag_amd
from verbatim_rag_amd.vector_stores import DenseShard
n, dim, k = 1_250_000, 768, 10
rng = np.random.default_rng(0)
sh = DenseShard(dim, n, "bf16")
for _ in range(n // 125_000):
    sh.add((rng.integers(-64, 65, size=(125_000, dim)) / 64.0).astype(np.float32))
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for nq, exact in ((33, True), (64, True), (128, True), (256, True), (1024, True), (32, False), (128, False)):
    q = (rng.integers(-64, 65, size=(nq, dim)) / 64.0).astype(np.float32) if exact else rng.standard_normal((nq, dim)).astype(np.float32)
    sh.search(q, k)
    dt = timed(lambda: sh.run_resident(nq, k))
    print(json.dumps({"tile64": os.environ.get("VRAG_TOPK_TILE64", "1"), "plan": os.environ.get("VRAG_TOPK_STAGE_PLAN", "1"), "rows": "bf16", "nq": nq, "exact_bf16_queries": exact, "ms": round(dt * 1e3, 3),
                      "TBps": round(n * dim * 2 / dt / 1e12, 2)}))
sh.close()
