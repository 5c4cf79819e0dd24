"""GpuVectorStore.query_batch == per-query GpuVectorStore.query, through the C ABI (one batched device pass per method
vs one pass per query).  Data are unit-norm vectors with entries +-1/sqrt(dim) (bf16-exact, norm exactly 1) and dyadic
sparse weights, so every score is exact in both kernels and ties are frequent: the `(score desc, id asc)` order is what
is being compared."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _store(n, dim, vocab, dtype, seed):
    from verbatim_rag_amd.vector_stores import GpuVectorStore

    rng = np.random.default_rng(seed)
    dense = (rng.integers(0, 2, (n, dim)) * 2 - 1).astype(np.float32) / np.float32(np.sqrt(dim))
    sparse = [{int(t): float(v) for t, v in zip(rng.choice(vocab, 12, replace=False), rng.integers(1, 64, 12) / 64)} for _ in range(n)]
    st = GpuVectorStore(dense_dim=dim, sparse_vocab=vocab, dense_dtype=dtype)
    st.add_vectors([f"id{i}" for i in range(n)], dense.tolist(), sparse, [f"text {i}" for i in range(n)],
                   [f"enh {i}" for i in range(n)], [{"document_id": f"d{i % 3}", "n": i} for i in range(n)])
    return st, dense, sparse, rng


def _dump(per_q):
    return [[(r.id, r.score, r.text, r.metadata) for r in rs] for rs in per_q]


@pytest.mark.parametrize("n,dim,dtype", [(3000, 64, "f32"), (3000, 64, "bf16"), (150000, 1024, "bf16"), (700, 256, "bf16")])
def test_store_query_batch_equals_per_query(n, dim, dtype):
    vocab = 2000
    st, dense, sparse, rng = _store(n, dim, vocab, dtype, n + dim)
    nq = 37
    dq = [dense[int(i)].tolist() for i in rng.integers(0, n, nq)]
    dq[3] = ((rng.integers(0, 2, dim) * 2 - 1).astype(np.float32) / np.float32(np.sqrt(dim))).tolist()
    sq = [{int(t): float(v) for t, v in zip(rng.choice(vocab, 6, replace=False), rng.integers(1, 64, 6) / 64)} for _ in range(nq)]
    sq[5] = {vocab - 1: 1.0, 0: 0.25}
    tq = [f"q{i}" for i in range(nq)]
    cases = [dict(dense_queries=dq, search_type="dense", top_k=5), dict(sparse_queries=sq, search_type="sparse", top_k=7),
             dict(dense_queries=dq, sparse_queries=sq, search_type="hybrid", top_k=5),
             dict(dense_queries=dq, sparse_queries=sq, top_k=4, hybrid_weights={"dense": 0.7, "sparse": 0.3, "full_text": 2.0}, rrf_k=30),
             dict(dense_queries=dq, sparse_queries=sq, top_k=4, hybrid_weights={"sparse": 1.0}),
             dict(dense_queries=dq, search_type="dense", top_k=6, filter='metadata["document_id"] == "d1"'),
             dict(dense_queries=dq, sparse_queries=sq, search_type="hybrid", top_k=3, filter='metadata["n"] in [0, 1, 2, 3, 5, 8, 13, 21, 34]')]
    try:
        for round_ in range(2):
            for kw in cases:
                rest = {k: v for k, v in kw.items() if not k.endswith("_queries")}
                want = [st.query(dense_query=kw.get("dense_queries", [None] * nq)[i], sparse_query=kw.get("sparse_queries", [None] * nq)[i],
                                 text_query=tq[i], **rest) for i in range(nq)]
                got = st.query_batch(text_queries=tq, **kw)
                assert _dump(got) == _dump(want), (round_, rest)
            st.delete([f"id{i}" for i in range(0, n, 7)])       # second round: deleted rows eat top slots
    finally:
        if st._dense is not None:
            st._dense.close()
        for shard, _base, _n in st._sparse_parts:
            shard.close()


def test_store_query_batch_side_branches():
    st, dense, sparse, rng = _store(200, 64, 500, "f32", 5)
    assert [len(r) for r in st.query_batch(text_queries=["a", "b"], top_k=3)] == [3, 3]           # filter-only browse
    with pytest.raises(ValueError):
        st.query_batch(dense_queries=[dense[0].tolist()], search_type="bogus")
    with pytest.raises(ValueError):
        st.query_batch(dense_queries=[dense[0].tolist()], sparse_queries=[{}, {}], search_type="dense")
    mixed = st.query_batch(dense_queries=[dense[0].tolist(), None], sparse_queries=[sparse[0], sparse[1]], top_k=2,
                           hybrid_weights={"dense": 1.0, "sparse": 1.0})
    want = [st.query(dense_query=dense[0].tolist(), sparse_query=sparse[0], top_k=2, hybrid_weights={"dense": 1.0, "sparse": 1.0}),
            st.query(dense_query=None, sparse_query=sparse[1], top_k=2, hybrid_weights={"dense": 1.0, "sparse": 1.0})]
    assert _dump(mixed) == _dump(want)
    st._dense.close()
    for shard, _base, _n in st._sparse_parts:
        shard.close()
