"""Timeline subject: a few public store calls at 10^6 rows (dense fp32 single query, sparse 1 / 16 queries), each repeated; WHAT selects one."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import verbatim_rag_amd
from verbatim_rag_amd.vector_stores import DenseShard, SparseShard
what = os.environ.get("WHAT", "dense1")
rng = np.random.default_rng(0)
if what.startswith("dense"):
    n, dim, k = 1_000_000, 768, 10
    sh = DenseShard(dim, n, "f32")
    for _ in range(8): sh.add(rng.standard_normal((125_000, dim), dtype=np.float32))
    nq = int(what[5:])
    q = rng.standard_normal((nq, dim), dtype=np.float32)
    for _ in range(6): sh.search(q, k)
else:
    n, vocab, k = 1_000_000, 30522, 5
    nnz = np.maximum(1, rng.poisson(128, size=n)); indptr = np.zeros(n + 1, np.int64); np.cumsum(nnz, out=indptr[1:])
    p = 1.0 / np.arange(1, vocab + 1); p /= p.sum()
    idx = rng.choice(vocab, size=int(indptr[-1]), p=p).astype(np.int32)
    val = (rng.integers(1, 193, size=int(indptr[-1])) / 64.0).astype(np.float32)
    sh = SparseShard(vocab, indptr, idx, val)
    nq = int(what[6:])
    qs = [{int(t): float(v) for t, v in zip(rng.choice(vocab, 32, p=p), rng.integers(1, 193, 32) / 64.0)} for _ in range(nq)]
    for _ in range(6): sh.search(qs, k)
torch.cuda.synchronize()
sh.close()
