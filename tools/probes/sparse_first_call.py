"""First-call vs warmed cost of the batched sparse search (10^6 docs, 1 000 queries, k = 5) through the public call."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
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
t0 = time.perf_counter(); sh = SparseShard(vocab, indptr, idx, val); torch.cuda.synchronize(); t_build = time.perf_counter() - t0
qs = [{int(t): float(v) for t, v in zip(rng.choice(vocab, 32, p=p), rng.integers(1, 193, 32) / 64.0)} for _ in range(1000)]
out = {"build_s": round(t_build, 3)}
for name, q in (("first_1000", qs), ("second_1000", qs), ("third_16", qs[:16]), ("fourth_1000", qs)):
    t0 = time.perf_counter(); sh.search(q, k); out[name + "_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
print(json.dumps(out))
sh.close()
