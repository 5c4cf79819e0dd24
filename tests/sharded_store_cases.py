"""Shared by the sharded-store tests (CPU gloo stand-ins and the GPU run): one scripted session against a
`GpuVectorStore` -- inserts in several batches (one smaller than the world size), deletes, dense / sparse / hybrid /
weighted / filtered queries, single and batched -- whose full transcript must not depend on how the rows are sharded."""
import contextlib

import numpy as np

import verbatim_rag_amd  # noqa: F401
from verbatim_rag_amd import vector_stores as vs


class CpuDense:
    """Stand-in for `DenseShard` on boxes without a GPU: answers from the exact CPU oracle (tests only)."""

    def __init__(self, dim, capacity, dtype="f32", device=0, prefilter=True):
        self.rows = np.zeros((0, dim), np.float32)

    def add(self, rows):
        self.rows = np.concatenate([self.rows, np.asarray(rows, np.float32)])

    def search(self, queries, k, stream=None):
        from oracle import topk_ref as T

        kk = min(k, len(self.rows))
        s, i = T.dense_topk(self.rows, np.asarray(queries, np.float32), kk)
        pad = k - kk
        return np.pad(s, ((0, 0), (0, pad))), np.pad(i, ((0, 0), (0, pad)), constant_values=-1)

    def close(self):
        pass


class CpuSparse:
    def __init__(self, vocab, indptr, indices, values, device=0):
        self.vocab, self.csr = vocab, (indptr, indices, values)

    def search(self, queries, k, stream=None):
        from oracle import topk_ref as T

        n = len(self.csr[0]) - 1
        kk = min(k, n)
        s, i = T.sparse_topk(*self.csr, self.vocab, *vs.dicts_to_csr(list(queries)), kk)
        pad = k - kk
        return np.pad(s, ((0, 0), (0, pad))), np.pad(i, ((0, 0), (0, pad)), constant_values=-1)

    def close(self):
        pass


@contextlib.contextmanager
def cpu_stand_ins():
    from verbatim_rag_amd.distributed import merge_topk

    saved = (vs._lib.load, vs._lib.require_gpu, vs.DenseShard, vs.SparseShard, vs._merge_parts)
    vs._lib.load, vs._lib.require_gpu, vs.DenseShard, vs.SparseShard = (lambda: None), (lambda: None), CpuDense, CpuSparse
    vs._merge_parts = lambda scores, rows, k, device: merge_topk(scores, rows, k)      # segments of one shard: host statement
    try:
        yield
    finally:
        vs._lib.load, vs._lib.require_gpu, vs.DenseShard, vs.SparseShard, vs._merge_parts = saved


def _dump(results):
    return [(r.id, float(r.score), r.text, r.enhanced_text, sorted(r.metadata.items())) for r in results]


def build_and_query(comm=None, dense_dtype="f32", dim=64, vocab=300, n=403, seed=11):
    """Dyadic-grid data (sums exact in any order, so scores do not depend on the kernel's summation order)."""
    rng = np.random.default_rng(seed)
    dense = (rng.integers(-8, 9, (n, dim)) / 8).astype(np.float32)
    dense[:, 0] = 1.0                                                     # no zero rows
    # rows are normalised inside the store; make every norm a power of two so the unit rows stay dyadic
    dense[:, 1:] = np.where(rng.random((n, dim - 1)) < 0.5, 0.5, -0.5).astype(np.float32)
    dense[:, 0] = 0.5                                                     # |row|^2 = dim / 4 = 16 -> norm 4
    sparse = [{int(t): float(v) for t, v in zip(rng.choice(vocab, 9, replace=False), rng.integers(1, 64, 9) / 64)} for _ in range(n)]
    st = vs.GpuVectorStore(dense_dim=dim, sparse_vocab=vocab, dense_dtype=dense_dtype, comm=comm)
    cuts = [0, 1, 2, 150, 151, n]                                         # batches of 1 (fewer rows than ranks), 148, 1, ...
    for a, b in zip(cuts[:-1], cuts[1:]):
        st.add_vectors([f"id{i}" for i in range(a, b)], dense[a:b].tolist(), sparse[a:b], [f"text {i}" for i in range(a, b)],
                       [f"enh {i}" for i in range(a, b)], [{"document_id": f"d{i % 3}", "n": i} for i in range(a, b)])
    out = []
    nq = 9
    picks = rng.integers(0, n, nq)
    dq = [dense[int(i)].tolist() for i in picks]
    sq = [{int(t): float(v) for t, v in zip(rng.choice(vocab, 5, replace=False), rng.integers(1, 64, 5) / 64)} for _ in range(nq)]
    cases = [dict(dense_queries=dq, search_type="dense", top_k=5), dict(sparse_queries=sq, search_type="sparse", top_k=7),
             dict(dense_queries=dq, sparse_queries=sq, search_type="hybrid", top_k=5),
             dict(dense_queries=dq, sparse_queries=sq, top_k=4, hybrid_weights={"dense": 0.7, "sparse": 0.3}, rrf_k=30),
             dict(dense_queries=dq, search_type="dense", top_k=6, filter='metadata["document_id"] == "d1"'),
             dict(dense_queries=dq, sparse_queries=sq, search_type="hybrid", top_k=3, filter='metadata["n"] in [0, 1, 2, 3, 5, 8, 9, 11]'),
             dict(dense_queries=dq, search_type="dense", top_k=70)]        # more than one device page
    for round_ in range(2):
        for kw in cases:
            rest = {k: v for k, v in kw.items() if not k.endswith("_queries")}
            batch = st.query_batch(text_queries=[f"q{i}" for i in range(nq)], **kw)
            out.append([_dump(r) for r in batch])
            one = st.query(dense_query=kw.get("dense_queries", [None] * nq)[0], sparse_query=kw.get("sparse_queries", [None] * nq)[0],
                           text_query="q0", **rest)
            assert _dump(one) == _dump(batch[0]), (round_, rest)
        st.delete([f"id{i}" for i in range(0, n, 3)])
    st.add_vectors(["late"], [dense[7].tolist()], [sparse[7]], ["late text"], ["late enh"], [{"document_id": "d9", "n": 9999}])
    out.append(_dump(st.query(dense_query=dense[7].tolist(), top_k=3, search_type="dense")))
    out.append(_dump(st.query(dense_query=dense[7].tolist(), top_k=3, search_type="dense", filter='metadata["document_id"] == "d9"')))
    return out
