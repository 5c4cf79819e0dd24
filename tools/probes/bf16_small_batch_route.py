"""bf16 rows, 2-32 queries (exact bf16 and generic fp32): device-resident search time on whatever route the build takes (run once per build:
the product thresholds kTiledMinBf16 / kTiledMinBf16Pairs against a build with both set to 2)."""
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
out = {}
for exact in (True, False):
    for nq in (1, 2, 3, 4, 5, 8, 12, 16, 17, 24, 32, 33):
        q = (rng.integers(-64, 65, size=(nq, dim)) / 64.0).astype(np.float32) if exact else rng.standard_normal((nq, dim)).astype(np.float32)
        sh.search(q, k)
        out[f"{'exact' if exact else 'generic'}_{nq}"] = round(timed(lambda: sh.run_resident(nq, k)) * 1e3, 3)
print(json.dumps(out))
sh.close()
