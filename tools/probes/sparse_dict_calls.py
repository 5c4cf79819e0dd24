"""Per-call host-inclusive time of the first calls of SparseShard.search (dict queries) against search_csr at 16 queries."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import verbatim_rag_amd
from verbatim_rag_amd.vector_stores import SparseShard, dicts_to_csr
n, vocab, k = 1_000_000, 30522, 5
rng = np.random.default_rng(1)
nnz = np.maximum(1, rng.poisson(128, size=n)); indptr = np.zeros(n + 1, np.int64); np.cumsum(nnz, out=indptr[1:])
p = 1.0 / np.arange(1, vocab + 1); p /= p.sum()
idx = rng.choice(vocab, size=int(indptr[-1]), p=p).astype(np.int32)
val = (rng.integers(1, 193, size=int(indptr[-1])) / 64.0).astype(np.float32)
sh = SparseShard(vocab, indptr, idx, val)
out = {}
for nq in (16, 8, 2):
    qs = [{int(t): float(v) for t, v in zip(rng.choice(vocab, 32, p=p), rng.integers(1, 193, 32) / 64.0)} for _ in range(nq)]
    csr = dicts_to_csr(qs)
    def one(fn):
        torch.cuda.synchronize(); a = time.perf_counter(); fn(); torch.cuda.synchronize(); return round((time.perf_counter() - a) * 1e3, 3)
    out[nq] = {"dicts": [one(lambda: sh.search(qs, k)) for _ in range(12)], "csr": [one(lambda: sh.search_csr(*csr, k)) for _ in range(12)],
               "dicts_again": [one(lambda: sh.search(qs, k)) for _ in range(6)], "to_csr": [one(lambda: dicts_to_csr(qs)) for _ in range(6)]}
print(json.dumps(out))
sh.close()
