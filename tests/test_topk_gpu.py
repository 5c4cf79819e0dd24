"""Dense / sparse exact top-k on the GPU vs the C oracle (bit-exact indices on dyadic-grid data)."""
import numpy as np
import pytest

from oracle import topk_ref as T

pytestmark = pytest.mark.gpu


def _dyadic(rng, shape, lim=64):
    return rng.integers(-lim, lim + 1, size=shape).astype(np.float32) / 64.0


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("n,dim,nq,k", [(10000, 768, 5, 10), (4097, 384, 1, 5), (37, 64, 3, 64), (200000, 128, 9, 7)])
def test_dense_topk_exact_on_dyadic_grid(dtype, n, dim, nq, k):
    from verbatim_rag_amd.vector_stores import DenseShard

    rng = np.random.default_rng(n + dim)
    X, Q = _dyadic(rng, (n, dim)), _dyadic(rng, (nq, dim))
    X[n // 2] = X[n // 3]                        # exact score ties -> id asc must decide
    sh = DenseShard(dim, n + 5, dtype)
    sh.add(X[: n // 2])
    sh.add(X[n // 2:])
    s, i = sh.search(Q, k)
    rs, ri = T.dense_topk(X, Q, k)
    sh.close()
    assert np.array_equal(i, ri)
    assert np.array_equal(s, rs)


def _assert_same_ranking(got_s, got_i, ref_s, ref_i, rows, queries, tol=2e-6, max_swapped=0.02):
    """Top-k lists agree up to fp32 near-ties: wherever the ids differ, the two rows' exact (float64) scores are within
    `tol` of each other -- no fp32 implementation with a different summation order than the oracle's can do better --
    and such positions are rare."""
    assert got_i.shape == ref_i.shape and np.abs(got_s - ref_s).max() < tol
    diff = got_i != ref_i
    assert diff.mean() <= max_swapped, f"{diff.sum()} of {diff.size} positions differ"
    r64, q64 = rows.astype(np.float64), queries.astype(np.float64)
    for q, p in zip(*np.nonzero(diff)):
        a = float(r64[got_i[q, p]] @ q64[q])
        b = float(r64[ref_i[q, p]] @ q64[q])
        assert abs(a - b) < tol, (q, p, got_i[q, p], ref_i[q, p], a, b)
        assert set(got_i[q]) == set(ref_i[q]) or abs(a - float(ref_s[q, -1])) < tol


def test_dense_topk_random_data_ranking_equals_the_fp32_query_oracle():
    """Non-dyadic data (normalised Gaussian rows, fp32 queries that are NOT bf16 numbers).  The accumulation order
    differs from the oracle's sequential fmaf chain, so scores agree to fp32 summation noise; the top-k ids must be the
    oracle's, position by position, except where two rows' scores tie to within that noise (a handful of positions
    among thousands; dyadic-grid data, where every sum is exact, is compared bit for bit in the tests above).
    Single queries run the scalar fp32-query kernel; batches of >= 3 over bf16 rows run on the matrix cores with every
    query carried as a (bf16, bf16 remainder) column pair (topk.hip `split`: 16 significant bits -- with plain bf16
    queries 11 of these 400 positions come out different and scores are off by 2e-4); fp32 rows take the scalar
    kernels at every batch size."""
    from verbatim_rag_amd.vector_stores import DenseShard

    rng = np.random.default_rng(1)
    X = rng.standard_normal((60000, 768)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = X[:40] + 0.05 * rng.standard_normal((40, 768)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    assert (Q.view(np.uint32) & 0xFFFF).any()                    # genuinely fp32 queries
    for dtype, rows in (("bf16", T.bf16_round(X)), ("f32", X)):  # the oracle sees the rows the index stores
        sh = DenseShard(768, 60000, dtype)
        sh.add(X)
        sb, ib = sh.search(Q, 10)            # batched path
        s1, i1 = sh.search(Q[:1], 10)        # single-query path
        sh.close()
        rs, ri = T.dense_topk(rows, Q, 10)   # fp32 queries, sequential fp32 fmaf chain
        _assert_same_ranking(sb, ib, rs, ri, rows, Q)
        _assert_same_ranking(s1, i1, rs[:1], ri[:1], rows, Q[:1], max_swapped=0.2)
        assert (ib[:, 0] == np.arange(40)).all()
    # 384-d rows (the other matrix-core instantiation) and a dimension served by the first-generation kernel
    for dim in (384, 256):
        Xd = rng.standard_normal((30000, dim)).astype(np.float32)
        Xd /= np.linalg.norm(Xd, axis=1, keepdims=True)
        Qd = rng.standard_normal((19, dim)).astype(np.float32)
        Qd /= np.linalg.norm(Qd, axis=1, keepdims=True)
        sh = DenseShard(dim, 30000, "bf16")
        sh.add(Xd)
        sb, ib = sh.search(Qd, 8)
        sh.close()
        rows = T.bf16_round(Xd)
        rs, ri = T.dense_topk(rows, Qd, 8)
        _assert_same_ranking(sb, ib, rs, ri, rows, Qd)


@pytest.mark.parametrize("prefilter", [True, False])
def test_dense_f32_rows_are_bit_exact_on_arbitrary_data(prefilter):
    """prefilter=True (the default): 1, 2 and 70 queries rank the bf16 image of the rows for candidates and re-score them
    with the exact chain (csrc/topk.hip "fp32 rows, bf16 prefilter") -- the same bits as the full scan, which serves 5 and 33.
    fp32 rows (the store's default, like the reference's FLOAT_VECTOR field) run on v_mfma_f32_32x32x2_f32, whose
    result IS the oracle's sequential `acc = fmaf(x[c], q[c], acc)` chain: scores and ids equal oracle/topk_ref.c bit for
    bit on data where summation order matters (normalised Gaussian rows and queries), for every batch size -- one
    query, a partial pass, several passes -- and with ties broken by id (duplicated rows)."""
    from verbatim_rag_amd.vector_stores import DenseShard

    rng = np.random.default_rng(7)
    for n, dim in ((70001, 768), (33000, 384), (5000, 96)):
        X = rng.standard_normal((n, dim)).astype(np.float32)
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        X[n // 2] = X[3]                       # an exact duplicate far away: equal scores, the lower id first
        sh = DenseShard(dim, n, "f32", prefilter=prefilter)
        sh.add(X[: n // 3])
        sh.add(X[n // 3:])                     # appended in two calls
        for nq in (1, 2, 3, 4, 5, 8, 9, 33, 70, 300):     # 2-4 at dim 768: ONE image pass for the batch (prefilter_collect_multi_kernel, round 6); 5-256: collect form of the tiled search over the image; 300: its 64-candidate form
            Q = rng.standard_normal((nq, dim)).astype(np.float32)
            Q[0] = X[3] * np.float32(1.7)      # query aligned with the duplicated row
            for k in (1, 10, 16):
                s, i = sh.search(Q, k)
                rs, ri = T.dense_topk(X, Q, k)
                assert np.array_equal(i, ri), (n, dim, nq, k)
                assert np.array_equal(s, rs), (n, dim, nq, k, float(np.abs(s - rs).max()))
            if nq == 1:
                assert i[0, 0] == 3 and (k == 1 or i[0, 1] == n // 2)
        sh.close()


def test_dense_fewer_rows_than_k_and_empty():
    from verbatim_rag_amd.vector_stores import DenseShard

    sh = DenseShard(64, 100, "f32")
    s, i = sh.search(np.ones((2, 64), np.float32), 5)
    assert (i == -1).all() and np.isinf(s).all()
    sh.add(np.eye(3, 64, dtype=np.float32))
    s, i = sh.search(np.eye(1, 64, dtype=np.float32), 5)
    sh.close()
    assert i.tolist() == [[0, 1, 2, -1, -1]] and s[0, :3].tolist() == [1.0, 0.0, 0.0]


def _sparse_corpus(rng, n, vocab, mean_nnz, qn, q_nnz):
    p = 1.0 / np.arange(1, vocab + 1)
    p /= p.sum()
    indptr, idx, val = [0], [], []
    for _ in range(n):
        m = max(0, int(rng.poisson(mean_nnz)))
        t = np.unique(rng.choice(vocab, size=m, p=p)) if m else np.zeros(0, np.int64)
        idx.append(t)
        val.append(rng.integers(1, 193, size=len(t)).astype(np.float32) / 64)
        indptr.append(indptr[-1] + len(t))
    qp, qi, qv = [0], [], []
    for _ in range(qn):
        t = np.unique(rng.choice(vocab, size=max(1, int(rng.poisson(q_nnz))), p=p))
        qi.append(t)
        qv.append(rng.integers(1, 193, size=len(t)).astype(np.float32) / 64)
        qp.append(qp[-1] + len(t))
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    return (np.asarray(indptr), cat(idx, np.int32), cat(val, np.float32), np.asarray(qp), cat(qi, np.int32), cat(qv, np.float32))


@pytest.mark.parametrize("vocab", [30522, 50368])       # LDS-resident query / L2-resident query paths
def test_sparse_topk_bit_exact(vocab):
    from verbatim_rag_amd.vector_stores import SparseShard

    rng = np.random.default_rng(vocab)
    indptr, idx, val, qp, qi, qv = _sparse_corpus(rng, 20000, vocab, 64, 6, 24)
    sh = SparseShard(vocab, indptr, idx, val)
    st = sh.stats()
    assert st["n_docs"] == 20000 and st["nnz"] == indptr[-1] and st["padded_nnz"] < 1.1 * st["nnz"] + 64 * 64
    s, i = sh.search_csr(qp, qi, qv, 5)
    rs, ri = T.sparse_topk(indptr, idx, val, vocab, qp, qi, qv, 5)
    assert np.array_equal(i, ri) and np.array_equal(s, rs)
    # arbitrary fp32 weights: same fmaf order as the oracle -> still bit-exact
    val2 = rng.random(len(val)).astype(np.float32) * 3
    qv2 = rng.random(len(qv)).astype(np.float32) * 3
    sh2 = SparseShard(vocab, indptr, idx, val2)
    s, i = sh2.search_csr(qp, qi, qv2, 10)
    rs, ri = T.sparse_topk(indptr, idx, val2, vocab, qp, qi, qv2, 10)
    sh.close(); sh2.close()
    assert np.array_equal(i, ri) and np.array_equal(s, rs)


def test_sparse_query_with_a_repeated_term_keeps_the_last_weight():
    """A query CSR may name a term twice: the LAST weight counts (what a dense scatter of the pairs gives) -- on the
    single-query kernels, whose dense query vector is scattered on the device since round 6, and on the batched pass's
    host-built weight table alike.  Unsorted terms inside a query row are fine too (the document's term order is what counts)."""
    from verbatim_rag_amd.vector_stores import SparseShard

    vocab = 30522
    rng = np.random.default_rng(5)
    indptr, idx, val, _qp, _qi, _qv = _sparse_corpus(rng, 20000, vocab, 64, 1, 8)
    hot = np.bincount(idx, minlength=vocab).argsort()[-6:][::-1].astype(np.int32)     # frequent terms: many documents score
    # query 0: term hot[0] three times (weights 9, 0.25, 2 -> 2 counts), hot[1] once, hot[2] twice (-> 0.5); query 1: ordinary
    q0_t = np.asarray([hot[0], hot[1], hot[0], hot[2], hot[0], hot[2]], np.int32)
    q0_w = np.asarray([9.0, 1.5, 0.25, 3.0, 2.0, 0.5], np.float32)
    q1_t, q1_w = hot[3:6].copy(), np.asarray([1.0, 0.75, 2.5], np.float32)
    d0_t, d0_w = np.asarray([hot[0], hot[1], hot[2]], np.int32), np.asarray([2.0, 1.5, 0.5], np.float32)   # the de-duplicated query 0
    sh = SparseShard(vocab, indptr, idx, val)
    try:
        one = lambda t, w: (np.asarray([0, len(t)], np.int64), t, w)
        order0, order1 = np.argsort(d0_t), np.argsort(q1_t)
        r0 = T.sparse_topk(indptr, idx, val, vocab, *one(d0_t[order0], d0_w[order0]), 7)
        r1 = T.sparse_topk(indptr, idx, val, vocab, *one(q1_t[order1], q1_w[order1]), 7)
        s, i = sh.search_csr(*one(q0_t, q0_w), 7)                                   # single-query kernel
        assert np.array_equal(i, r0[1]) and np.array_equal(s, r0[0])
        qp = np.asarray([0, len(q0_t), len(q0_t) + len(q1_t)], np.int64)
        s, i = sh.search_csr(qp, np.concatenate([q0_t, q1_t]), np.concatenate([q0_w, q1_w]), 7)   # batched pass
        assert np.array_equal(i[0], r0[1][0]) and np.array_equal(s[0], r0[0][0])
        assert np.array_equal(i[1], r1[1][0]) and np.array_equal(s[1], r1[0][0])
    finally:
        sh.close()


@pytest.mark.parametrize("vocab", [30522, 50368])
def test_sparse_batched_hash_path_and_fallbacks_bit_exact(vocab):
    """>= 2 small queries: QB = 8 queries per pass through per-query hash tables in LDS (19 queries = 3 passes, the
    last one partial); one query alone and any batch containing a > 64-term query take the single-query kernels.
    All three must equal the oracle bit for bit (same fmaf order)."""
    from verbatim_rag_amd.vector_stores import SparseShard

    rng = np.random.default_rng(vocab + 1)
    indptr, idx, val, qp, qi, qv = _sparse_corpus(rng, 30000, vocab, 96, 19, 40)
    val = (rng.random(len(val)).astype(np.float32) * 3)
    qv = (rng.random(len(qv)).astype(np.float32) * 3)
    assert (np.diff(qp) <= 64).all() and (np.diff(qp) > 30).any()
    sh = SparseShard(vocab, indptr, idx, val)
    try:
        rs, ri = T.sparse_topk(indptr, idx, val, vocab, qp, qi, qv, 7)
        s, i = sh.search_csr(qp, qi, qv, 7)                         # batched hash path
        assert np.array_equal(i, ri) and np.array_equal(s, rs)
        s1, i1 = sh.search_csr(qp[:2], qi[:qp[1]], qv[:qp[1]], 7)   # one query: single-query kernel
        assert np.array_equal(i1, ri[:1]) and np.array_equal(s1, rs[:1])
        # a 100-term query in the batch: whole batch falls back to the single-query kernels
        big_t = np.unique(rng.choice(vocab, size=140))[:100].astype(np.int32)
        big_v = (rng.random(len(big_t)).astype(np.float32) * 2)
        qp2 = np.concatenate([qp, [qp[-1] + len(big_t)]])
        qi2, qv2 = np.concatenate([qi, big_t]), np.concatenate([qv, big_v])
        rs2, ri2 = T.sparse_topk(indptr, idx, val, vocab, qp2, qi2, qv2, 7)
        s2, i2 = sh.search_csr(qp2, qi2, qv2, 7)
        assert np.array_equal(i2, ri2) and np.array_equal(s2, rs2)
    finally:
        sh.close()


def test_sparse_no_shared_terms_is_not_a_hit():
    from verbatim_rag_amd.vector_stores import SparseShard

    sh = SparseShard(100, [0, 2, 3, 3], [1, 2, 5], [1.0, 2.0, 3.0])     # doc 2 is empty
    s, i = sh.search([{5: 2.0}, {7: 1.0}], 3)
    sh.close()
    assert i.tolist() == [[1, -1, -1], [-1, -1, -1]] and s[0, 0] == 6.0


def test_vector_store_semantics():
    from verbatim_rag_amd.vector_stores import GpuVectorStore

    rng = np.random.default_rng(3)
    n, dim, vocab = 300, 64, 1000
    dense = _dyadic(rng, (n, dim)).tolist()
    sparse = [{int(t): float(v) for t, v in zip(rng.choice(vocab, 8, replace=False), rng.integers(1, 64, 8) / 64)} for _ in range(n)]
    st = GpuVectorStore(dense_dim=dim, sparse_vocab=vocab, dense_dtype="f32")
    st.add_vectors([f"id{i}" for i in range(n)], dense, sparse, [f"text {i}" for i in range(n)],
                   [f"enh {i}" for i in range(n)], [{"document_id": f"d{i % 3}", "title": f"T{i}"} for i in range(n)])
    q = dense[17]
    r = st.query(dense_query=q, top_k=3, search_type="dense")
    assert r[0].id == "id17" and abs(r[0].score - 1.0) < 1e-5 and r[0].text == "text 17" and r[0].metadata["title"] == "T17"
    r = st.query(sparse_query=sparse[5], top_k=4, search_type="sparse")
    assert r[0].id == "id5" and len(r) <= 4
    r = st.query(dense_query=q, sparse_query=sparse[17], top_k=5, search_type="hybrid")
    assert r[0].id == "id17" and abs(r[0].score - (1.0 - (0.5 / 61 + 0.5 / 61))) < 1e-12     # RRF distance, float64
    r = st.query(dense_query=q, top_k=5, search_type="dense", filter='metadata["document_id"] == "d0"')
    assert all(x.metadata["document_id"] == "d0" for x in r) and len(r) == 5
    with pytest.raises(ValueError):
        st.query(dense_query=q, top_k=5, search_type="dense", filter='title like "T1%"')       # outside the supported subset
    with pytest.raises(ValueError):
        st.query(dense_query=q, top_k=5, search_type="bogus")
    st.delete(["id17"])
    assert st.query(dense_query=q, top_k=3, search_type="dense")[0].id != "id17"
    r = st.query(dense_query=q, sparse_query=sparse[3], top_k=4, hybrid_weights={"dense": 0.7, "sparse": 0.3, "full_text": 1})
    assert len(r) == 4
    assert len(st.query(top_k=7)) == 7     # filter-only browse
    # persistence round trip: same hits (ids, scores, text, metadata); the deleted row stays deleted
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        st.save(d)
        st2 = GpuVectorStore.load(d)
        for kw in (dict(dense_query=q, top_k=5, search_type="dense"), dict(sparse_query=sparse[5], top_k=4, search_type="sparse"),
                   dict(dense_query=q, sparse_query=sparse[9], top_k=6, search_type="hybrid"),
                   dict(dense_query=q, top_k=5, search_type="dense", filter='metadata["document_id"] == "d1"')):
            a, b = st.query(**kw), st2.query(**kw)
            assert [(x.id, x.score, x.text, x.metadata) for x in a] == [(x.id, x.score, x.text, x.metadata) for x in b]
        assert all(x.id != "id17" for x in st2.query(top_k=400))


@pytest.mark.parametrize("n,dim,nq,k", [(10000, 768, 40, 10), (5000, 128, 8, 16), (130, 256, 33, 5), (300000, 384, 64, 10),
                                         (200000, 768, 33, 16), (140000, 1024, 5, 3)])
def test_dense_topk_batched_mfma_path_exact(n, dim, nq, k):
    """>= 8 queries, bf16 rows, k <= 16: 32 queries per pass on the matrix cores (queries rounded to bf16;
    dyadic-grid queries are exact in bf16, so indices and scores must equal the oracle bit for bit).  Shards of
    >= 131 072 rows at dim 384 / 768 / 1024 take the two-pass route (prefix pass seeds the entry thresholds)."""
    from verbatim_rag_amd.vector_stores import DenseShard

    rng = np.random.default_rng(n + nq)
    X, Q = _dyadic(rng, (n, dim)), _dyadic(rng, (nq, dim))
    X[n // 2] = X[n // 3]
    Q[1] = Q[0]
    sh = DenseShard(dim, n, "bf16")
    sh.add(X)
    s, i = sh.search(Q, k)
    sh.close()
    rs, ri = T.dense_topk(X, Q, k)
    assert np.array_equal(i, ri)
    assert np.array_equal(s, rs)


@pytest.mark.parametrize("seed", range(8))
def test_dense_topk_fuzz_shapes(seed):
    """Random shard sizes around the tile / range boundaries of every dense path (single-query register path, 4-query
    LDS path, both MFMA generations, prefix-seeded two-pass route), random k and query counts, heavy score ties."""
    from verbatim_rag_amd.vector_stores import DenseShard

    rng = np.random.default_rng(7000 + seed)
    dim = int(rng.choice([64, 128, 256, 384, 768, 1024]))
    n = int(rng.choice([1, 31, 127, 128, 129, 2047, 2048, 2049, 32767, 32768, 65537, 131072, 131073, 150001]))
    if n * dim > 80_000_000:
        n = 80_000_000 // dim
    nq = int(rng.choice([1, 2, 3, 4, 5, 31, 32, 33, 40]))
    k = int(rng.choice([1, 2, 5, 10, 16, 17, 32]))
    dtype = "bf16" if rng.random() < 0.75 else "f32"
    X, Q = _dyadic(rng, (n, dim), lim=int(rng.choice([2, 8, 64]))), _dyadic(rng, (nq, dim), lim=int(rng.choice([2, 64])))
    sh = DenseShard(dim, n, dtype)
    sh.add(X)
    s, i = sh.search(Q, k)
    sh.close()
    rs, ri = T.dense_topk(X, Q, k)
    assert np.array_equal(i, ri), (n, dim, nq, k, dtype)
    assert np.array_equal(s, rs), (n, dim, nq, k, dtype)


def test_sparse_ties_order_by_document_id():
    """Short documents and queries with weights in {0.5, 1}: most hits tie.  The index stores documents sorted by
    length, so the key must carry the caller's document index for `(score desc, id asc)` to hold."""
    from verbatim_rag_amd.vector_stores import SparseShard

    rng = np.random.default_rng(77)
    n, vocab = 30000, 300
    lens = rng.integers(1, 9, n)
    indptr = np.concatenate([[0], np.cumsum(lens)])
    idx = np.concatenate([np.sort(rng.choice(vocab, int(m), replace=False)) for m in lens]).astype(np.int32)
    val = (rng.integers(1, 3, len(idx)) / 2).astype(np.float32)
    qp = np.arange(0, 3 * 21, 3)
    qi = np.concatenate([np.sort(rng.choice(vocab, 3, replace=False)) for _ in range(20)]).astype(np.int32)
    qv = (rng.integers(1, 3, len(qi)) / 2).astype(np.float32)
    sh = SparseShard(vocab, indptr, idx, val)
    try:
        for k in (5, 17, 64):
            rs, ri = T.sparse_topk(indptr, idx, val, vocab, qp, qi, qv, k)
            s, i = sh.search_csr(qp, qi, qv, k)                                   # batched kernel
            assert np.array_equal(i, ri) and np.array_equal(s, rs), k
            s1, i1 = sh.search_csr(qp[:2], qi[:3], qv[:3], k)                     # single-query kernel
            assert np.array_equal(i1, ri[:1]) and np.array_equal(s1, rs[:1]), k
    finally:
        sh.close()


@pytest.mark.parametrize("dtype,dim,n,nq,k", [("f32", 64, 5000, 5, 100), ("bf16", 768, 20000, 3, 200), ("bf16", 128, 90, 2, 100),
                                               ("f32", 256, 3000, 1, 1000), ("bf16", 384, 70000, 9, 65)])
def test_dense_paged_topk_beyond_64(dtype, dim, n, nq, k):
    """k > 64: exact pages of 64.  Rows are duplicated in runs so that page boundaries fall inside groups of equal
    scores (the `(score desc, id asc)` order has to carry across pages); n < k leaves -1 / -inf tails."""
    from verbatim_rag_amd.vector_stores import DenseShard

    rng = np.random.default_rng(n + k)
    base = _dyadic(rng, (max(1, n // 40), dim))
    X = base[rng.integers(0, len(base), n)]                   # ~40 copies of every distinct row
    Q = _dyadic(rng, (nq, dim))
    sh = DenseShard(dim, n, dtype)
    sh.add(X)
    s, i = sh.search(Q, k)
    s2, i2 = sh.search(Q[:1], 5)                              # a single-pass call afterwards still works
    sh.close()
    rs, ri = T.dense_topk(X, Q, min(k, n))
    assert np.array_equal(i[:, :min(k, n)], ri) and np.array_equal(s[:, :min(k, n)], rs)
    if n < k:
        assert (i[:, n:] == -1).all() and np.isneginf(s[:, n:]).all()
    assert np.array_equal(i2, ri[:1, :5]) and np.array_equal(s2, rs[:1, :5])


def test_sparse_paged_topk_beyond_64():
    from verbatim_rag_amd.vector_stores import SparseShard

    rng = np.random.default_rng(5)
    indptr, idx, val, qp, qi, qv = _sparse_corpus(rng, 20000, 3000, 12, 7, 6)
    val = (rng.integers(1, 5, len(val)) / 4).astype(np.float32)          # heavy ties
    qv = (rng.integers(1, 3, len(qv)) / 2).astype(np.float32)
    sh = SparseShard(3000, indptr, idx, val)
    try:
        for k in (65, 150, 1000):
            rs, ri = T.sparse_topk(indptr, idx, val, 3000, qp, qi, qv, k)
            s, i = sh.search_csr(qp, qi, qv, k)
            assert np.array_equal(i, ri) and np.array_equal(s, rs), k
        rs, ri = T.sparse_topk(indptr, idx, val, 3000, qp, qi, qv, 8)        # back to the batched single pass
        s, i = sh.search_csr(qp, qi, qv, 8)
        assert np.array_equal(i, ri) and np.array_equal(s, rs)
        with pytest.raises(Exception):
            sh.search_csr(qp, qi, qv, 1025)
    finally:
        sh.close()


def test_store_returns_more_than_64_hits():
    from verbatim_rag_amd.vector_stores import GpuVectorStore

    rng = np.random.default_rng(8)
    n, dim = 500, 64
    dense = (rng.integers(0, 2, (n, dim)) * 2 - 1).astype(np.float32) / np.float32(8.0)
    st = GpuVectorStore(dense_dim=dim, enable_sparse=False, sparse_vocab=None, dense_dtype="f32")
    st.add_vectors([f"id{i}" for i in range(n)], dense.tolist(), None, [f"t{i}" for i in range(n)], [""] * n,
                   [{"document_id": f"d{i % 2}"} for i in range(n)])
    r = st.query(dense_query=dense[3].tolist(), top_k=120, search_type="dense")
    rs, ri = T.dense_topk(dense, dense[3:4], 120)
    assert [x.id for x in r] == [f"id{j}" for j in ri[0]]
    r = st.query(dense_query=dense[3].tolist(), top_k=100, search_type="dense", filter='metadata["document_id"] == "d1"')
    odd = np.arange(1, n, 2)
    rs, ri = T.dense_topk(dense[odd], dense[3:4], 100)
    assert [x.id for x in r] == [f"id{odd[j]}" for j in ri[0]]
    st._dense.close()


def test_sparse_edges_max_vocab_empty_queries_and_bad_terms():
    """u16 term ids: the largest vocabulary is 65 536 (term 65 535 usable); the batched kernel needs vocab <= 65 535
    and hands such a shard to the single-query kernel.  Queries without terms / without matches give -1 rows."""
    from verbatim_rag_amd._lib import VragError
    from verbatim_rag_amd.vector_stores import SparseShard

    V = 65536
    indptr = [0, 2, 3, 3, 5]                              # doc 2 is empty
    idx = [0, 65535, 65535, 7, 65535]
    val = [1.0, 2.0, 0.5, 1.0, 0.25]
    sh = SparseShard(V, indptr, idx, val)
    try:
        s, i = sh.search([{65535: 2.0}, {}, {12345: 1.0}, {0: 1.0, 7: 3.0}], 4)
        rs, ri = T.sparse_topk(np.asarray(indptr), np.asarray(idx, np.int32), np.asarray(val, np.float32), V,
                               np.asarray([0, 1, 1, 2, 4]), np.asarray([65535, 12345, 0, 7], np.int32),
                               np.asarray([2.0, 1.0, 1.0, 3.0], np.float32), 4)
        assert np.array_equal(i, ri) and np.array_equal(s, rs)
        assert i[0].tolist() == [0, 1, 3, -1] and i[1].tolist() == [-1] * 4 and i[2].tolist() == [-1] * 4 and i[3].tolist() == [3, 0, -1, -1]
        with pytest.raises(VragError):
            sh.search([{65536: 1.0}], 1)
        with pytest.raises(VragError):
            sh.search([{-1: 1.0}], 1)
    finally:
        sh.close()
    with pytest.raises(VragError):
        SparseShard(V + 1, [0, 1], [0], [1.0])
    with pytest.raises(VragError):
        SparseShard(100, [0, 1], [100], [1.0])            # document term outside the vocabulary


def test_dense_edges_dim_limits_and_capacity():
    from verbatim_rag_amd._lib import VragError
    from verbatim_rag_amd.vector_stores import DenseShard

    rng = np.random.default_rng(4)
    for dim, dtype in ((8, "f32"), (8, "bf16"), (4096, "bf16")):
        X, Q = _dyadic(rng, (70, dim)), _dyadic(rng, (3, dim))
        sh = DenseShard(dim, 70, dtype)
        sh.add(X[:33])
        sh.add(X[33:])                                     # incremental adds
        s, i = sh.search(Q, 7)
        rs, ri = T.dense_topk(X, Q, 7)
        assert np.array_equal(i, ri) and np.array_equal(s, rs), (dim, dtype)
        with pytest.raises(VragError):
            sh.add(X[:1])                                  # beyond capacity
        sh.close()
    for bad in (12, 0, 4104):
        with pytest.raises(VragError):
            DenseShard(bad, 10, "bf16")
    sh = DenseShard(64, 10, "bf16")                        # empty shard: every slot is a miss
    s, i = sh.search(_dyadic(rng, (2, 64)), 3)
    sh.close()
    assert (i == -1).all() and np.isneginf(s).all()


@pytest.mark.parametrize("n,dim,nq,k", [(70_000, 768, 64, 10), (300_000, 384, 300, 5), (5_000, 128, 100, 16), (200_000, 1024, 257, 64),
                                         (4_096, 64, 64, 1), (4_353, 256, 65, 7), (1_100_000, 64, 500, 33),
                                         # the first-stage forms of the stage plan (late round 6): a whole round of keys + one appending stage;
                                         # lists longer than 16 with n mod round inside the key budget (first stage = 5 000 rows); 1 100 queries
                                         # (column tiles that do not divide the grid, 512-row first stage)
                                         (140_000, 64, 40, 10), (136_072, 64, 257, 40), (20_000, 64, 1100, 5)])
def test_dense_topk_tiled_batched_search_exact(n, dim, nq, k):
    """>= 64 queries over >= 4 096 bf16 rows: the shard is read once per batch -- rows x queries on the encoder's GEMM
    kernel, EPI_TOPK epilogue, staged thresholds (csrc/topk.hip).  Dyadic-grid data: ids and scores equal the oracle bit
    for bit, ties (duplicated rows and queries, a small value range) included; row counts around the stage boundaries
    256 / 4 096 / 65 536 / 1 048 576 and query counts off the 256-column tile."""
    from verbatim_rag_amd.vector_stores import DenseShard

    rng = np.random.default_rng(n + nq)
    X, Q = _dyadic(rng, (n, dim), lim=8 if dim <= 128 else 64), _dyadic(rng, (nq, dim))
    X[n // 2] = X[n // 3]
    X[n - 1] = X[0]
    Q[1] = Q[0]
    sh = DenseShard(dim, n, "bf16")
    sh.add(X)
    s, i = sh.search(Q, k)
    s2, i2 = sh.search(Q[:63], k)          # 64 query columns: the 256 x 64 tile form (the 32-queries-per-pass route under VRAG_TOPK_NO_TILED)
    sh.close()
    rs, ri = T.dense_topk(X, Q, k, blocked=n * nq > 5_000_000)
    assert np.array_equal(i, ri), (n, dim, nq, k)
    assert np.array_equal(s, rs), (n, dim, nq, k)
    assert np.array_equal(i2, ri[:63]) and np.array_equal(s2, rs[:63])


def test_dense_topk_tiled_stage_plan_around_its_boundaries():
    """The stage plan of the tiled search (csrc/topk.hip dense_tiled_search) branches on the shard size against the tile round
    (65 536 / 32 768 / 16 384 rows for <= 256 / 512 / 1 024 query columns), on n mod round against the first stage's least size
    and key budget, on the list length (<= 16: 16x growth, longer: 4x and an 8 192-row first stage at most) and on the batch
    (widened last stage up to 256 queries): seeded sizes ON and next to those boundaries, every one bit for bit the oracle."""
    from verbatim_rag_amd.vector_stores import DenseShard

    R = 65536
    cases = [(R, 64, 10), (R + 1, 64, 10), (R - 1, 33, 3), (R + 4095, 64, 10), (R + 4096, 64, 16), (R + 4097, 40, 17),
             (2 * R, 128, 10), (2 * R + 5000, 200, 10), (2 * R + 5000, 256, 16), (3 * R // 2, 257, 10), (R + 4500, 300, 20),
             (32768 + 4096, 512, 5), (16384 * 3 + 100, 1000, 10), (8192 + 4200, 70, 64), (40_000, 520, 33), (4097, 64, 10),
             # >= 4 rounds of rows and <= 256 query columns: the first stage is a SAMPLE of the shard's tiles (every (tiles / 256)-th),
             # its keys are dropped after the selection and the appending stages cover all rows -- at the switch, off a tile multiple,
             # with a sampling period of 2, and (the last case, see below) with the rows in topic order
             (4 * R - 256, 64, 10), (4 * R, 64, 10), (4 * R + 300, 200, 16), (8 * R + 77, 40, 10), (6 * R + 5, 2, 10)]
    for n, nq, k in cases:
        rng = np.random.default_rng(n * 7 + nq)
        X, Q = _dyadic(rng, (n, 64), lim=8), _dyadic(rng, (nq, 64))
        X[n - 1] = X[0]                         # a tie across the whole shard
        if n == 6 * R + 5:                      # topic order: 5 000 consecutive rows far better for query 0 than the rest
            X[200_000:205_000, :8] = 4.0 + rng.integers(0, 8, size=(5_000, 8)).astype(np.float32) / 8     # bf16-exact
            Q[0] = 0
            Q[0, :8] = 1.0
        sh = DenseShard(64, n, "bf16")
        sh.add(X)
        s, i = sh.search(Q, k)
        sh.close()
        rs, ri = T.dense_topk(X, Q, k, blocked=True)
        assert np.array_equal(i, ri) and np.array_equal(s, rs), (n, nq, k)
    # fp32 rows behind the bf16 image: the collect form's prefix (min(n, 65 536) rows) as one direct stage or 4 096 + the rest,
    # the 64-candidate route above 256 queries -- arbitrary data, the exact fmaf chain's bits
    for n, nq, k in [(R - 7, 5, 10), (R, 33, 10), (R + 4100, 64, 16), (R + 4100, 100, 10), (2 * R + 300, 256, 5), (R + 9000, 300, 10)]:
        rng = np.random.default_rng(n + nq)
        X, Q = rng.standard_normal((n, 64), dtype=np.float32), rng.standard_normal((nq, 64), dtype=np.float32)
        sh = DenseShard(64, n, "f32")
        sh.add(X)
        s, i = sh.search(Q, k)
        sh.close()
        rs, ri = T.dense_topk(X, Q, k, blocked=True)
        assert np.array_equal(i, ri) and np.array_equal(s, rs), ("f32", n, nq, k)


def test_dense_pass_kernels_in_a_child_process():
    """Since late round 6 a bf16 shard of >= 4 096 rows with dim % 64 == 0 answers every batch of two or more queries through the
    tiled search; the 4- / 32-queries-per-pass kernels (scalar and matrix-core, one- and two-pass plans) keep the other shards
    and `VRAG_TOPK_NO_TILED` (read once per process).  A child process runs this module's dense tests under it, so the pass
    kernels stay pinned to the oracle at the full range of shapes."""
    import os
    import subprocess
    import sys

    if os.environ.get("VRAG_TOPK_NO_TILED"):
        pytest.skip("already inside the child")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_topk_gpu.py"), "-q", "-x", "-m", "gpu", "-k",
         "batched_mfma_path or exact_on_dyadic_grid or random_data_ranking or fuzz_shapes or tiled_batched_search_exact or fp32_queries_ride or dense_paged"],
        env={**os.environ, "VRAG_TOPK_NO_TILED": "1"}, capture_output=True, text=True, timeout=1500, cwd=root)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_dense_topk_tiled_fp32_queries_ride_as_column_pairs():
    """Queries that are not bf16 numbers: (value, remainder) column pairs in the tiled search too -- the ranking equals the
    fp32-query oracle's up to fp32 summation noise, exactly like the 16-queries-per-pass route."""
    from verbatim_rag_amd.vector_stores import DenseShard

    rng = np.random.default_rng(12)
    X = rng.standard_normal((90_000, 768)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = X[:150] + 0.05 * rng.standard_normal((150, 768)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    assert (Q.view(np.uint32) & 0xFFFF).any()
    rows = T.bf16_round(X)
    sh = DenseShard(768, len(X), "bf16")
    sh.add(X)
    sb, ib = sh.search(Q, 10)
    s64, i64 = sh.search(Q[:64], 10)          # 128 query columns: the 256 x 128 tile form
    sh.close()
    rs, ri = T.dense_topk(rows, Q, 10, blocked=True)
    _assert_same_ranking(sb, ib, rs, ri, rows, Q)
    _assert_same_ranking(s64, i64, rs[:64], ri[:64], rows, Q[:64])
    assert (ib[:, 0] == np.arange(150)).all()


def test_dense_topk_tiled_candidate_overflow_is_rescued():
    """Rows sorted by similarity to a query: every row of a stage beats the threshold the stage started with, the
    query's candidate buffer (2 048 slots) overflows, and the query is re-answered from the whole shard -- by the pass kernels behind
    the flags the host call reads back, by the sliced rescue pass where the lists stay on the device.
    Query 0 scores row r as x0 + x1 / 256 with (x0, x1) ascending in r -- 16 641 distinct, strictly increasing scores;
    query 1 sees them descending; the other queries are ordinary."""
    from verbatim_rag_amd.vector_stores import DenseShard

    rng = np.random.default_rng(77)
    n, dim, nq, k = 16_641, 128, 70, 10
    X = _dyadic(rng, (n, dim), lim=4)
    r = np.arange(n)
    X[:, 0] = ((r // 129) - 64) / 64.0
    X[:, 1] = ((r % 129) - 64) / 64.0
    Q = _dyadic(rng, (nq, dim))
    Q[0] = 0
    Q[0, 0], Q[0, 1] = 1.0, 1.0 / 256
    Q[1] = -Q[0]
    import torch

    sh = DenseShard(dim, n, "bf16")
    sh.add(X)
    s, i = sh.search(Q, k)                     # the host call: flags read back with the lists, flagged queries through the pass kernels
    d_s = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    d_i = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    sh.search_device(Q, k, d_s.data_ptr(), d_i.data_ptr(), stream=None)     # no host round trip: the device rescue pass (sliced walk + merge)
    s2, i2 = sh.search(Q[:2], k)               # the two flagged queries alone (64 query columns)
    torch.cuda.synchronize()
    sh.close()
    rs, ri = T.dense_topk(X, Q, k)
    assert np.array_equal(i[0], np.arange(n - 1, n - 1 - k, -1)) and np.array_equal(i[1], np.arange(k))
    assert np.array_equal(i, ri) and np.array_equal(s, rs)
    assert np.array_equal(d_i.cpu().numpy(), ri) and np.array_equal(d_s.cpu().numpy(), rs)
    assert np.array_equal(i2, ri[:2]) and np.array_equal(s2, rs[:2])


def test_dense_f32_prefilter_falls_back_when_scores_bunch():
    """300 copies of one row (and 300 near-copies, 1e-4 apart -- inside the image's error bound): the 64 best approximate
    scores cannot be separated from the rest, the sufficiency test fails and the full fp32 scan answers -- ids ascending among
    the exact ties, near-copies ranked by their exact scores.  Un-normalised rows far from the query keep the bound large."""
    from verbatim_rag_amd.vector_stores import DenseShard

    rng = np.random.default_rng(31)
    n, dim = 20_000, 256
    X = rng.standard_normal((n, dim)).astype(np.float32)
    base = X[7].copy()
    dup = rng.choice(np.arange(100, n), size=600, replace=False)
    X[dup[:300]] = base
    X[dup[300:]] = base + (1e-4 * rng.standard_normal((300, dim))).astype(np.float32)
    Q = rng.standard_normal((70, dim)).astype(np.float32)
    Q[0] = base
    sh = DenseShard(dim, n, "f32")
    sh.add(X)
    for qs in (Q[:1], Q[:2], Q[:3], Q[:4], Q[:7], Q[:8], Q):     # dim 256: two to four queries share one image pass (query 0 overflows its list there too); 7, 8: tiled
        s, i = sh.search(qs, 16)
        rs, ri = T.dense_topk(X, qs, 16)
        assert np.array_equal(i, ri) and np.array_equal(s, rs), len(qs)
    sh.close()
    assert set(ri[0][:8].tolist()) <= set([7] + dup.tolist())


def test_sparse_batched_passes_over_a_large_shard():
    """Batches of >= 32 queries (16 per pass) over 9 375 slices, one workgroup per CU: (a) ordinary data; (b) one-term documents
    whose weight FALLS with the document number, so that the documents a wave sees first are the best ones and the lists are
    filled once and then only compared against (the register-resident insertion path's quiet case) -- both bit-exact."""
    from verbatim_rag_amd.vector_stores import SparseShard, dicts_to_csr

    rng = np.random.default_rng(5)
    n, vocab, k = 600_000, 1000, 5
    # (a) 4-12 terms per document, weights on the dyadic grid
    lens = rng.integers(4, 13, n)
    ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ix = rng.integers(0, vocab, int(ip[-1])).astype(np.int32)
    vv = (rng.integers(1, 193, len(ix)) / 64.0).astype(np.float32)
    qs = [{int(t): float(w) for t, w in zip(rng.choice(vocab, 6, replace=False), rng.integers(1, 64, 6) / 64)} for _ in range(40)]
    sp = SparseShard(vocab, ip, ix, vv)
    s, i = sp.search(qs, k)
    sp.close()
    rs, ri = T.sparse_topk(ip, ix, vv, vocab, *dicts_to_csr(qs), k, blocked=True)
    assert np.array_equal(i, ri) and np.array_equal(s, rs)
    # (b) falling weights
    ip = np.arange(n + 1, dtype=np.int64)
    ix = (np.arange(n) % 7).astype(np.int32)
    vv = (2.0 - np.arange(n, dtype=np.float64) / n).astype(np.float32)
    qs = [{int(q % 7): 1.0, int(7 + q): 0.5} for q in range(33)]
    sp = SparseShard(vocab, ip, ix, vv)
    s, i = sp.search(qs, k)
    sp.close()
    rs, ri = T.sparse_topk(ip, ix, vv, vocab, *dicts_to_csr(qs), k, blocked=True)
    assert np.array_equal(i, ri) and np.array_equal(s, rs)
    assert np.array_equal(i[3], 3 + 7 * np.arange(k))


def test_dense_rows_ingested_from_device_memory_equal_host_ingest():
    """vrag_dense_index_add_device (embeddings that never visit the host): the same shard, both row dtypes, the fp32 one with its
    prefilter image built from the device rows."""
    import torch

    from verbatim_rag_amd.vector_stores import DenseShard

    rng = np.random.default_rng(3)
    X = rng.standard_normal((9_000, 384)).astype(np.float32)
    Q = rng.standard_normal((70, 384)).astype(np.float32)
    t = torch.from_numpy(X).cuda()
    for dtype in ("f32", "bf16"):
        a, b = DenseShard(384, len(X), dtype), DenseShard(384, len(X), dtype)
        a.add(X)
        b.add_device(t[:5000].data_ptr(), 5000)
        b.add_device(t[5000:].contiguous().data_ptr(), 4000)
        assert len(b) == len(a) == 9000
        for qs in (Q[:1], Q[:5], Q):
            sa, ia = a.search(qs, 10)
            sb, ib = b.search(qs, 10)
            assert np.array_equal(ia, ib) and np.array_equal(sa, sb), (dtype, len(qs))
        a.close()
        b.close()


def test_device_resident_search_of_fp32_rows_takes_the_prefilter_route_and_its_gated_fallback():
    """vrag_dense_index_search_device on fp32 rows with the prefilter image, >= 64 queries (what a rank of the sharded store runs per
    batch): no host round trip, so queries whose candidates fail the sufficiency test are re-answered by the full scan launched behind
    their flags.  Equal to the host-synchronous search (itself equal to the oracle) on ordinary data and on bunched scores."""
    import torch

    from verbatim_rag_amd.vector_stores import DenseShard

    rng = np.random.default_rng(41)
    n, dim, k = 30_000, 384, 10
    for bunched in (False, True):
        X = rng.standard_normal((n, dim)).astype(np.float32)
        Q = rng.standard_normal((70, dim)).astype(np.float32)
        if bunched:
            dup = rng.choice(np.arange(50, n), size=400, replace=False)
            X[dup[:200]] = X[7]
            X[dup[200:]] = X[7] + (1e-4 * rng.standard_normal((200, dim))).astype(np.float32)
            Q[:5] = X[7] * np.linspace(0.5, 2.0, 5, dtype=np.float32)[:, None]     # five queries flagged, the rest not
        sh = DenseShard(dim, n, "f32")
        sh.add(X)
        hs, hi = sh.search(Q, k)
        rs, ri = T.dense_topk(X, Q, k, blocked=True)
        assert np.array_equal(hi, ri) and np.array_equal(hs, rs)
        d_s = torch.empty((70, k), dtype=torch.float32, device="cuda")
        d_i = torch.empty((70, k), dtype=torch.int64, device="cuda")
        sh.search_device(Q, k, d_s.data_ptr(), d_i.data_ptr(), id_base=1000, stream=None)
        torch.cuda.synchronize()
        sh.close()
        assert np.array_equal(d_i.cpu().numpy(), ri + 1000), bunched
        assert np.array_equal(d_s.cpu().numpy(), rs), bunched


def test_one_and_two_queries_over_fp32_rows_take_the_one_pass_route_host_and_device():
    """1-8 queries (round 6: two to four share ONE pass where dim % 256 == 0; the name is round 5's) over fp32 rows with the prefilter image (csrc/topk.hip prefilter_single_enqueue): entry threshold from the best
    keys of the prefix workgroups, ONE pass over the image for the candidates, exact chains out of LDS, one selection.  Through the
    host call and through vrag_dense_index_search_device (gated full scan behind the overflow flag), on ordinary rows and on a
    shard with 5 000 copies of the best row -- more candidates than the list holds, so the full scan answers; ties by id."""
    import torch

    from verbatim_rag_amd.vector_stores import DenseShard

    rng = np.random.default_rng(53)
    for n, dim in ((40_000, 768), (9_001, 224)):     # 224: no register-resident query form; % 32: the exact full scan
        for copies in (0, 5000):
            X = rng.standard_normal((n, dim)).astype(np.float32)
            Q = rng.standard_normal((8, dim)).astype(np.float32)
            if copies:
                X[rng.choice(np.arange(10, n), size=copies, replace=False)] = X[7]
                Q[0] = X[7]
                Q[3] = X[7] * np.float32(0.5)
            sh = DenseShard(dim, n, "f32")
            sh.add(X)
            for nq in (1, 2, 3, 4, 6, 8):    # dim 768: 2-4 queries in one pass (queries 0 and 3 overflow their candidate lists when rows are copied); 6, 8: tiled
                for k in (1, 10, 16):
                    rs, ri = T.dense_topk(X, Q[:nq], k)
                    s, i = sh.search(Q[:nq], k)
                    assert np.array_equal(i, ri) and np.array_equal(s, rs), (n, dim, copies, nq, k)
                    d_s = torch.empty((nq, k), dtype=torch.float32, device="cuda")
                    d_i = torch.empty((nq, k), dtype=torch.int64, device="cuda")
                    sh.search_device(Q[:nq], k, d_s.data_ptr(), d_i.data_ptr(), id_base=500, stream=None)
                    torch.cuda.synchronize()
                    assert np.array_equal(d_i.cpu().numpy(), ri + 500), (n, dim, copies, nq, k)
                    assert np.array_equal(d_s.cpu().numpy(), rs), (n, dim, copies, nq, k)
            sh.close()
            if copies:
                assert ri[0, 0] == 7 and (np.diff(ri[0]) > 0).all()     # the copies tie: ascending ids


def test_prefilter_bound_holds_where_bf16_rounding_is_worst_case():
    """The image's error bound is the MEASURED max ||bf16(x) - x|| (csrc/topk.hip prefilter_image_kernel / prefilter_eps), not a
    constant: bf16 keeps 8 significant bits, so a row of halfway values (1 + 2^-8 -> 1.0) loses 2^-8 of its norm, all of it along
    the query.  Ten such rows T (exact score S (1 + 2^-8), image score S) sit behind the ranked prefix; forty decoys in the prefix
    use the other half of the coordinates with values just above a halfway point of the binade below (image 1.0, exact 1 - 2^-9)
    and a query weight 1.485 * 2^-8 higher: image scores ABOVE T's by 1.485 * 2^-8 S, exact scores BELOW T's.  A bound of 2^-9 per
    element (what the first form of the route assumed) puts T outside t0 - 2 eps and returns the decoys; the measured bound keeps T."""
    import torch

    from verbatim_rag_amd.vector_stores import DenseShard

    rng = np.random.default_rng(61)
    n, dim, k = 50_000, 128, 10       # 128: the accumulation slack of the bound (4 dim 2^-24) is small against the rounding term
    h = dim // 2
    X = (0.01 * rng.standard_normal((n, dim))).astype(np.float32)
    decoys = rng.choice(np.arange(100, 30_000), size=40, replace=False)
    truth = np.sort(rng.choice(np.arange(40_000, n), size=k, replace=False))
    X[decoys] = 0.0
    X[decoys, h:] = np.float32(1.0 - 2.0 ** -9 + 2.0 ** -20)
    X[truth] = 0.0
    X[truth, :h] = np.float32(1.0 + 2.0 ** -8)
    Q = np.ones((2, dim), dtype=np.float32)
    Q[0, h:] = np.float32(1.0 + 1.485 * 2.0 ** -8)
    Q[1] = rng.standard_normal(dim).astype(np.float32)
    rs, ri = T.dense_topk(X, Q, k)
    assert np.array_equal(ri[0], truth)                       # the construction: exact top-k = the T rows, ids ascending
    img = X.astype(np.float32).view(np.uint32)
    img = (((img + 0x7FFF + ((img >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)).view(np.float32)   # bf16 image, round to nearest even
    a = img.astype(np.float64) @ Q[0].astype(np.float64)
    old_eps = (2.0 ** -9 + 2.0 ** -16 + 4 * dim * 2.0 ** -24) * 1.001 * np.linalg.norm(X, axis=1).max() * np.linalg.norm(Q[0])
    assert a[decoys].min() - a[truth].max() > 2 * old_eps     # ... and the 2^-9 bound (with all its slack) would have dropped them
    sh = DenseShard(dim, n, "f32")
    sh.add(X)
    for nq in (1, 2):
        s, i = sh.search(Q[:nq], k)
        assert np.array_equal(i, ri[:nq]) and np.array_equal(s, rs[:nq]), nq
        d_s = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        d_i = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        sh.search_device(Q[:nq], k, d_s.data_ptr(), d_i.data_ptr(), stream=None)
        torch.cuda.synchronize()
        assert np.array_equal(d_i.cpu().numpy(), ri[:nq]) and np.array_equal(d_s.cpu().numpy(), rs[:nq]), nq
    Qb = np.repeat(Q[:1], 70, axis=0) * np.linspace(0.5, 2.0, 70, dtype=np.float32)[:, None]      # the batch route on the same rows
    s, i = sh.search(Qb, k)
    rs, ri = T.dense_topk(X, Qb, k, blocked=True)
    assert np.array_equal(i, ri) and np.array_equal(s, rs)
    sh.close()


def test_device_bf16_conversion_rounds_to_nearest_even():
    """`prefilter_eps` (csrc/topk.hip) measures ||bf16(q) - q|| on the HOST for the batch route's rounded queries, assuming the
    device's `(bf16_t)` cast rounds to nearest even like its own bit arithmetic.  Pinned here on the cast as the bf16 ingest kernel
    compiles it: one value per row, query = e_0, so the returned score IS the stored bf16 value."""
    from verbatim_rag_amd.vector_stores import DenseShard

    vals = np.array([1 + 2.0 ** -8, 1 + 3 * 2.0 ** -8, 1 + 2.0 ** -8 + 2.0 ** -20, 1 + 2.0 ** -8 - 2.0 ** -20, -(1 + 2.0 ** -8),
                     1 - 2.0 ** -9, 1 - 2.0 ** -9 + 2.0 ** -20, 1 - 2.0 ** -9 - 2.0 ** -20, 3.0e38, 1.0e-30, 0.1, -0.3, 255.5, 257.0,
                     65535.0, 1.99609375 + 2.0 ** -9], dtype=np.float32)
    bits = vals.view(np.uint32).astype(np.uint64)
    want = (((bits + 0x7FFF + ((bits >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)).view(np.float32)
    X = np.zeros((len(vals), 8), dtype=np.float32)
    X[:, 0] = vals
    q = np.zeros((1, 8), dtype=np.float32)
    q[0, 0] = 1.0
    sh = DenseShard(8, len(vals), "bf16")
    sh.add(X)
    s, i = sh.search(q, len(vals))
    sh.close()
    got = np.empty(len(vals), dtype=np.float32)
    got[i[0]] = s[0]
    assert np.array_equal(got, want), (got, want)
