import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import verbatim_rag_amd
from verbatim_rag_amd.vector_stores import DenseShard
n, dim, k = 1_250_000, 768, 10
rng = np.random.default_rng(0)
sh = DenseShard(dim, n, "f32")
for _ in range(n // 125_000):
    sh.add(rng.standard_normal((125_000, dim)).astype(np.float32))
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for nq in (3, 4, 8, 16, 32, 48, 63, 64, 128):
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    dt = timed(lambda: sh.search(q, k))
    print(json.dumps({"pf_min": os.environ.get("VRAG_PF_MIN_BATCH", "64"), "nq": nq, "ms": round(dt * 1e3, 3)}))
sh.close()
