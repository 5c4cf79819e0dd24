import json, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/verbatim_rag_amd.py") else os.getcwd())
import torch, verbatim_rag_amd
from verbatim_rag_amd.vector_stores import DenseShard
n, dim, k = 1_250_000, 768, 10
rng = np.random.default_rng(0)
sh = DenseShard(dim, n, "f32")
for _ in range(n // 125_000): sh.add(rng.standard_normal((125_000, dim)).astype(np.float32))
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for nq in (256, 512, 1024):
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    print(json.dumps({"nq": nq, "ms": round(timed(lambda: sh.search(q, k)) * 1e3, 3)}))
sh.close()
