"""Host-inclusive latency of ONE GpuVectorStore.query call (10^6 rows, 768-d fp32 dense + SPLADE-shaped sparse): dense / sparse / hybrid."""
import json, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import verbatim_rag_amd
import synth_corpus as S
from verbatim_rag_amd.vector_stores import GpuVectorStore
n, dim, V = 1_000_000, 768, 30522
X = S.dense_rows(n, dim, seed=41)
ip, ix, vv = S.sparse_corpus(n, V, seed=21)
st = GpuVectorStore(dense_dim=dim, sparse_vocab=V)
st.add_vectors([f"c{i}" for i in range(n)], X, (ip, ix, vv), [f"chunk {i}" for i in range(n)], [f"chunk {i}" for i in range(n)], [{"n": i} for i in range(n)])
del X
Q = S.dense_rows(64, dim, seed=42); dq, _ = S.sparse_queries(64, V, seed=43)
def timed(fn, reps=64):
    for i in range(8): fn(i)
    t0 = time.perf_counter()
    for i in range(reps): fn(i)
    return (time.perf_counter() - t0) / reps * 1e3
out = {"dense_ms": timed(lambda i: st.query(dense_query=Q[i % 64].tolist(), top_k=5, search_type="dense")),
       "dense_ndarray_ms": timed(lambda i: st.query(dense_query=Q[i % 64], top_k=5, search_type="dense")),
       "sparse_ms": timed(lambda i: st.query(sparse_query=dq[i % 64], top_k=5, search_type="sparse")),
       "hybrid_ms": timed(lambda i: st.query(dense_query=Q[i % 64].tolist(), sparse_query=dq[i % 64], top_k=5, search_type="hybrid"))}
print(json.dumps({k: round(v, 3) for k, v in out.items()}))
