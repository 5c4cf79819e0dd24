"""Timeline subject: ONE batch size of the dense search, a few resident runs; MODE=bf16 (bf16 rows, tiled search) or f32 (fp32 rows + image, collect form)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import verbatim_rag_amd
from verbatim_rag_amd.vector_stores import DenseShard
n, dim, k = 1_250_000, 768, 10
nq = int(os.environ.get("NQ", "64"))
mode = os.environ.get("MODE", "bf16")
rng = np.random.default_rng(0)
sh = DenseShard(dim, n, mode)
for _ in range(n // 125_000):
    sh.add((rng.integers(-64, 65, size=(125_000, dim)) / 64.0).astype(np.float32) if mode == "bf16" else rng.standard_normal((125_000, dim), dtype=np.float32))
q = (rng.integers(-64, 65, size=(nq, dim)) / 64.0).astype(np.float32) if mode == "bf16" else rng.standard_normal((nq, dim), dtype=np.float32)
sh.search(q, k)
for _ in range(6): sh.run_resident(nq, k) if mode == "bf16" else sh.search(q, k)
torch.cuda.synchronize()
sh.close()
