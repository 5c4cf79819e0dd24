"""Kernel trace subject: the tiled dense search at one batch size (bf16 rows 1.25 M x 768, NQ exact bf16 queries, k = 10), 20 resident runs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import verbatim_rag_amd
from verbatim_rag_amd.vector_stores import DenseShard
n, dim, k = 1_250_000, 768, 10
nq = int(os.environ.get("NQ", "64"))
rng = np.random.default_rng(0)
sh = DenseShard(dim, n, "bf16")
for _ in range(n // 125_000):
    sh.add((rng.integers(-64, 65, size=(125_000, dim)) / 64.0).astype(np.float32))
q = (rng.integers(-64, 65, size=(nq, dim)) / 64.0).astype(np.float32)
sh.search(q, k)
for _ in range(20): sh.run_resident(nq, k)
torch.cuda.synchronize()
sh.close()
