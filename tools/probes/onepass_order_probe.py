"""Does the order of calls matter?  fp32 rows 1.25 M x 768, host search: each nq timed right after ingest in the order given (argv)."""
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
for a in sys.argv[1:]:
    nq = int(a)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    print(json.dumps({"nq": nq, "search_ms": round(timed(lambda: sh.search(q, k)) * 1e3, 3)}))
sh.close()
