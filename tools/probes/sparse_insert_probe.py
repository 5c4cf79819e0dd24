#!/usr/bin/env python3
"""Probe of the batched sparse pass (round 5): list-insertion forms x queries per pass on ONE resident 10^6-document shard.
VRAG_SPARSE_INSERT: 0 = ds_bpermute maximum, no gate (round-4 form); 1 = DPP maximum, no gate; 2 = workgroup gate + ds_bpermute
maximum; 3 = gate + DPP maximum.  Prints ms per 64-query search (resident queries) and per pass."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import verbatim_rag_amd  # noqa: E402,F401
from verbatim_rag_amd.vector_stores import SparseShard  # noqa: E402

n, vocab, k = 1_000_000, 30522, 5
rng = np.random.default_rng(1)
nnz = np.maximum(1, rng.poisson(128, size=n))
indptr = np.zeros(n + 1, np.int64)
np.cumsum(nnz, out=indptr[1:])
p = 1.0 / np.arange(1, vocab + 1)
p /= p.sum()
idx = rng.choice(vocab, size=int(indptr[-1]), p=p).astype(np.int32)
val = (rng.integers(1, 193, size=int(indptr[-1])) / 64.0).astype(np.float32)
sh = SparseShard(vocab, indptr, idx, val)
qs = [{int(t): float(v) for t, v in zip(rng.choice(vocab, 32, p=p), rng.integers(1, 193, 32) / 64.0)} for _ in range(64)]
refs = {}
for qb8, wgs, k in ((False, 256, 5), (False, 512, 5), (True, 256, 5), (False, 256, 10), (True, 256, 10)):
    if True:
        variant = "registers"
        os.environ["VRAG_SPARSE_WGS"] = str(wgs)
        if qb8:
            os.environ["VRAG_SPARSE_QB8"] = "1"
        else:
            os.environ.pop("VRAG_SPARSE_QB8", None)
        s, i = sh.search(qs, k)
        if k not in refs:
            refs[k] = (s, i)
        same = bool(np.array_equal(s, refs[k][0]) and np.array_equal(i, refs[k][1]))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            sh.run_resident(64, k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        per = 8 if qb8 else 16
        print(json.dumps({"queries_per_pass": per, "insert_variant": variant, "target_wgs": wgs, "k": k, "ms_64_queries": dt * 1e3, "ms_per_pass": dt * 1e3 / (64 // per),
                          "us_per_query": dt * 1e6 / 64, "same_result": same}), flush=True)
sh.close()
