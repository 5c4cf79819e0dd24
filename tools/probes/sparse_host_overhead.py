"""Where the host-inclusive time of a 1 000-query sparse search goes: dict -> CSR (Python), the C call (tables + uploads + kernels), kernels alone."""
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
for nq in (1, 16, 1000):
    qs = [{int(t): float(v) for t, v in zip(rng.choice(vocab, 32, p=p), rng.integers(1, 193, 32) / 64.0)} for _ in range(nq)]
    def t(fn, reps):
        fn(); torch.cuda.synchronize(); a = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - a) / reps * 1e3
    reps = 50 if nq < 1000 else 5
    csr = dicts_to_csr(qs)
    out[nq] = {"search_dicts_ms": round(t(lambda: sh.search(qs, k), reps), 3), "dicts_to_csr_ms": round(t(lambda: dicts_to_csr(qs), reps), 3),
               "search_csr_ms": round(t(lambda: sh.search_csr(*csr, k), reps), 3), "kernels_only_ms": round(t(lambda: sh.run_resident(nq, k), reps), 3)}
print(json.dumps(out))
sh.close()
