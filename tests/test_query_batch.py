"""Cross-query batched front end -- host logic on CPU: `HotPathIndex.query_batch` / `StaticVerbatimPipeline.query_batch`
must equal the per-query calls (whose dispatch is pinned against the reference by test_host_parity.py)."""
import types

import pytest

import verbatim_rag_amd  # noqa: F401
from verbatim_rag_amd.index import HotPathIndex
from verbatim_rag_amd.pipeline import StaticVerbatimPipeline
from verbatim_rag_amd.vector_stores import SearchResult, VectorStore


class Dense:
    def __init__(self):
        self.calls = []

    def embed_text(self, t):
        self.calls.append(("text", t))
        return [float(len(t)), 1.0]

    def embed_batch(self, ts):
        return [[float(len(t)), 1.0] for t in ts]

    def embed_queries(self, ts):
        self.calls.append(("queries", tuple(ts)))
        return [[float(len(t)), 1.0] for t in ts]

    def get_dimension(self):
        return 2


class SparseNoBatch:
    def embed_text(self, t):
        return {len(t): 0.5}

    def embed_batch(self, ts):
        return [{len(t): 0.5, 0: 1e-9} for t in ts]        # ingest rule differs from the query rule on purpose

    def get_dimension(self):
        return 100


class Store(VectorStore):
    enable_full_text = False

    def __init__(self, batched):
        self.calls = []
        if batched:
            self.query_batch = self._query_batch

    def add_vectors(self, *a, **k):
        pass

    def delete(self, ids):
        pass

    def query(self, dense_query=None, sparse_query=None, text_query=None, top_k=5, search_type="hybrid", filter=None,
              search_params=None, hybrid_weights=None, rrf_k=60):
        self.calls.append("query")
        if search_type == "dense" and not dense_query and hybrid_weights is None:
            raise ValueError("Invalid search configuration")
        key = (tuple(dense_query or ()), tuple(sorted((sparse_query or {}).items())), text_query, top_k, search_type, filter,
               tuple(sorted((hybrid_weights or {}).items())), rrf_k)
        return [SearchResult(id=f"{hash(key) % 1000}-{i}", score=1.0 - 0.1 * i, metadata={"title": "T", "source": "s"},
                             text=f"{text_query} hit {i}. Second sentence {i}.", enhanced_text="") for i in range(top_k)]

    def _query_batch(self, dense_queries=None, sparse_queries=None, text_queries=None, top_k=5, search_type="hybrid",
                     filter=None, search_params=None, hybrid_weights=None, rrf_k=60):
        self.calls.append("query_batch")
        n = len(text_queries)
        return [Store.query(self, dense_queries[i] if dense_queries else None, sparse_queries[i] if sparse_queries else None,
                            text_queries[i], top_k, search_type, filter, search_params, hybrid_weights, rrf_k) for i in range(n)]


def _dump(per_q):
    return [[(r.id, r.score, r.text) for r in rs] for rs in per_q]


CASES = [dict(), dict(search_type="dense"), dict(search_type="sparse"), dict(search_type="hybrid", k=3),
         dict(hybrid_weights={"dense": 0.7, "sparse": 0.3}), dict(hybrid_weights={"sparse": 1.0}, rrf_k=10),
         dict(filter='metadata["document_id"] == "d1"')]


@pytest.mark.parametrize("batched_store", [False, True])
@pytest.mark.parametrize("kw", CASES)
def test_index_query_batch_equals_per_query(batched_store, kw):
    texts = ["where is it?", "who?", "a much longer question about towers"]
    d, s = Dense(), SparseNoBatch()
    store = Store(batched_store)
    idx = HotPathIndex(store, dense_provider=d, sparse_provider=s)
    want = _dump([idx.query(t, **kw) for t in texts])
    store.calls.clear()
    d.calls.clear()
    assert _dump(idx.query_batch(texts, **kw)) == want
    if batched_store:
        assert store.calls.count("query_batch") == 1
        uses_dense = kw.get("search_type", "auto") in ("auto", "dense", "hybrid") and "dense" in kw.get("hybrid_weights", {"dense": 1})
        assert d.calls == ([("queries", tuple(texts))] if uses_dense else [])
    else:
        assert "query_batch" not in store.calls and len(store.calls) == len(texts)


def test_index_query_batch_side_branches():
    store = Store(True)
    idx = HotPathIndex(store, dense_provider=None, sparse_provider=SparseNoBatch())
    assert idx.query_batch([]) == []
    with pytest.raises(ValueError):                       # same error as query(): dense asked for, no dense provider
        idx.query_batch(["a", "b"], search_type="dense")
    store.calls.clear()
    out = idx.query_batch(["a", ""], k=2)                 # an empty text is the filter-only browse of query()
    assert store.calls.count("query_batch") == 0 and len(out) == 2


class Extractor:
    def __init__(self, batched):
        self.batches = 0
        if batched:
            self.extract_spans_batch = self._batch

    def extract_spans(self, question, results):
        return {r.text: [r.text.split(". ")[0] + "."] for r in results}

    def _batch(self, questions, per_q):
        self.batches += 1
        return [self.extract_spans(q, r) for q, r in zip(questions, per_q)]


@pytest.mark.parametrize("batched", [False, True])
def test_pipeline_query_batch_equals_per_query(batched):
    qs = ["where is it?", "who built it?", "when?"]
    ext = Extractor(batched)
    pipe = StaticVerbatimPipeline(HotPathIndex(Store(batched), dense_provider=Dense(), sparse_provider=SparseNoBatch()), ext, k=3)
    want = [pipe.query(q).model_dump() for q in qs]
    assert [r.model_dump() for r in pipe.query_batch(qs)] == want
    assert ext.batches == (1 if batched else 0)
    rer = types.SimpleNamespace(rerank=lambda q, rs: list(reversed(rs)))
    pipe = StaticVerbatimPipeline(pipe.index, ext, k=3, reranker=rer)
    assert [r.model_dump() for r in pipe.query_batch(qs)] == [pipe.query(q).model_dump() for q in qs]
