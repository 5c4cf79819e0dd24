"""Sparse batched top-k timing subject (10^6 docs, ~128 terms, vocab 30 522, Zipf terms; queries of 32 terms, k = 5): ms per search
at 16 / 64 / 1 000 resident queries.  A/B two library builds with VRAG_AMD_LIB."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import verbatim_rag_amd
from verbatim_rag_amd.vector_stores import SparseShard
n, vocab, mean_nnz, k = 1_000_000, 30522, 128, 5
rng = np.random.default_rng(1)
nnz = np.maximum(1, rng.poisson(mean_nnz, size=n))
indptr = np.zeros(n + 1, np.int64); np.cumsum(nnz, out=indptr[1:])
p = 1.0 / np.arange(1, vocab + 1); p /= p.sum()
idx = rng.choice(vocab, size=int(indptr[-1]), p=p).astype(np.int32)
val = (rng.integers(1, 193, size=int(indptr[-1])) / 64.0).astype(np.float32)
sh = SparseShard(vocab, indptr, idx, val)
def timed(fn, reps):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
ref = None
for nq in (16, 64, 1000):
    qs = [{int(t): float(v) for t, v in zip(rng.choice(vocab, 32, p=p), rng.integers(1, 193, 32) / 64.0)} for _ in range(nq)]
    s, i = sh.search(qs, k)
    dt = timed(lambda: sh.run_resident(nq, k), 10 if nq < 1000 else 3)
    print(json.dumps({"lib": os.path.basename(os.environ.get("VRAG_AMD_LIB", "libvrag_amd.so")), "nq": nq, "ms": round(dt * 1e3, 4),
                      "checksum": [float(np.asarray(s, np.float64).sum()), int(np.asarray(i, np.int64).sum())]}))
sh.close()
