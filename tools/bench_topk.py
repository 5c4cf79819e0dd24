#!/usr/bin/env python3
"""Top-k roofline probe (HBM-bound): dense bf16 rows and SELL-64 sparse rows, device-resident queries.
Prints one JSON object per index kind: rows/s, algorithmic GB/s, fraction of the 8 TB/s HBM peak."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (device sync only)

import verbatim_rag_amd  # noqa: E402,F401
from verbatim_rag_amd.vector_stores import DenseShard, SparseShard  # noqa: E402

HBM_PEAK = 8.0e12


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def dense(n=1_250_000, dim=768, k=10):
    rng = np.random.default_rng(0)
    sh = DenseShard(dim, n, "bf16")
    chunk = 125_000
    for _ in range(n // chunk):
        sh.add((rng.integers(-64, 65, size=(chunk, dim)) / 64.0).astype(np.float32))
    out = []
    tiled_off = os.environ.get("VRAG_TOPK_NO_TILED") is not None
    for nq, exact in ((1, True), (4, True), (32, True), (64, True), (256, True), (256, False), (1024, True), (10240, True)):
        if exact:   # bf16-exact queries: 32 per matrix-core pass
            q = (rng.integers(-64, 65, size=(nq, dim)) / 64.0).astype(np.float32)
        else:       # generic fp32 queries: (bf16, remainder) column pairs, 16 per pass
            q = rng.standard_normal((nq, dim)).astype(np.float32)
        sh.search(q, k)
        dt = timeit(lambda: sh.run_resident(nq, k), 10 if nq <= 1024 else 3)
        tiled = nq >= 64 and not tiled_off     # csrc/topk.hip dense_use_tiled: the shard is read once per batch
        per_pass = 1 if nq == 1 else (4 if nq < 3 else (32 if exact else 16))
        passes = 1 if tiled else (nq + per_pass - 1) // per_pass
        bytes_ = n * dim * 2 * passes          # one pass streams the shard once, whatever the number of queries in it
        flops = 2.0 * n * dim * nq * (1 if exact else 2)
        out.append({"kind": "dense_bf16", "rows": n, "dim": dim, "nq": nq, "k": k, "fp32_queries_split": not exact,
                    "route": "tiled_gemm_epilogue" if tiled else "query_passes",
                    "ms": dt * 1e3, "queries_per_s": nq / dt, "algorithmic_GBps": bytes_ / dt / 1e9,
                    "frac_of_8TBps": bytes_ / dt / HBM_PEAK, "bytes_per_pass": n * dim * 2, "passes": passes,
                    "TFLOPs": flops / dt / 1e12, "frac_of_2p5_PFLOPs": flops / dt / 2.5e15})
    sh.close()
    return out


def sparse(n=1_000_000, vocab=30522, mean_nnz=128, k=5):
    rng = np.random.default_rng(1)
    nnz = np.maximum(1, rng.poisson(mean_nnz, size=n))
    indptr = np.zeros(n + 1, np.int64)
    np.cumsum(nnz, out=indptr[1:])
    p = 1.0 / np.arange(1, vocab + 1)
    p /= p.sum()
    idx = rng.choice(vocab, size=int(indptr[-1]), p=p).astype(np.int32)   # duplicates inside a doc are harmless here
    val = (rng.integers(1, 193, size=int(indptr[-1])) / 64.0).astype(np.float32)
    sh = SparseShard(vocab, indptr, idx, val)
    st = sh.stats()
    out = []
    qb = 8 if vocab * 2 + 16 * 4096 + 16 * 16 * k * 8 > 160 * 1024 else 16   # csrc/topk.hip sparse_multi_fits
    for nq in (1, 8, 16, 64, 1000):
        qs = [{int(t): float(v) for t, v in zip(rng.choice(vocab, 32, p=p), rng.integers(1, 193, 32) / 64.0)} for _ in range(nq)]
        sh.search(qs, k)
        dt = timeit(lambda: sh.run_resident(nq, k), 5)
        per_pass = 1 if nq == 1 else (8 if nq <= 8 else qb)   # sparse_topk_kernel / sparse_topk_multi_kernel<8 | 16> (csrc/topk.hip sparse_pass_queries; up to eight queries take the 8-query pass)
        passes = (nq + per_pass - 1) // per_pass
        pass_bytes = st["padded_nnz"] * 6 + (st["n_docs"] // 64 + 1) * 12
        bytes_ = pass_bytes * passes            # a pass reads the shard ONCE for all of its queries
        out.append({"kind": "sparse_sell64x4", "docs": n, "nnz": st["nnz"], "padded_nnz": st["padded_nnz"], "nq": nq, "k": k,
                    "queries_per_pass": per_pass,
                    "ms": dt * 1e3, "queries_per_s": nq / dt, "algorithmic_GBps": bytes_ / dt / 1e9,
                    "frac_of_8TBps": bytes_ / dt / HBM_PEAK, "bytes_per_pass": pass_bytes, "passes": passes})
    sh.close()
    return out


if __name__ == "__main__":
    for row in dense() + sparse():
        print(json.dumps(row))
