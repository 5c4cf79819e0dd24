"""The drop-in claim, executed: the REFERENCE's own `VerbatimIndex` and `VerbatimRAG` (imported from /root/reference,
build container only -- skipped elsewhere) drive this package's store / providers / extractor, and the result equals
this package's restatement of the same pipeline (`HotPathIndex` + `StaticVerbatimPipeline`).  Device classes are
replaced by CPU stand-ins (oracle-backed shards, a logit function of the token ids): what is under test is every
Python-level seam between the two code bases -- argument names, return types, `SearchResult` fields, dict keys."""
import os
import sys
import types

import numpy as np
import pytest

import verbatim_rag_amd  # noqa: F401
from test_fast_packer import RecordingEngine
from test_store_host_logic import _Dense, _Sparse
from verbatim_rag_amd import vector_stores as vs

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture()
def reference(tmp_path, monkeypatch):
    stubs = tmp_path / "stubs"
    (stubs / "rapidfuzz").mkdir(parents=True)
    (stubs / "rapidfuzz" / "__init__.py").write_text("")
    (stubs / "rapidfuzz" / "fuzz.py").write_text("def partial_ratio_alignment(*a, **k):\n    raise RuntimeError('stub')\n")
    (stubs / "openai").mkdir()
    (stubs / "openai" / "__init__.py").write_text("class OpenAI:\n    def __init__(self, *a, **k): pass\nclass AsyncOpenAI(OpenAI):\n    pass\n")
    for p in (str(stubs), os.path.join(REF, "packages", "core"), REF):
        monkeypatch.syspath_prepend(p)
    from verbatim_rag.core import VerbatimRAG
    from verbatim_rag.index import VerbatimIndex

    return VerbatimIndex, VerbatimRAG


class Dense:
    def embed_text(self, t):
        v = np.zeros(64, np.float32)
        v[len(t) % 64] = 1.0
        v[(len(t) * 7) % 64] += 0.5
        return v.tolist()

    def embed_batch(self, ts):
        return [self.embed_text(t) for t in ts]

    def get_dimension(self):
        return 64


class Sparse:
    def embed_text(self, t):
        return {(len(w) * 13 + ord(w[0])) % 300: 1.0 + 0.25 * (len(w) % 3) for w in t.split()[:12]}

    def embed_batch(self, ts):
        return [self.embed_text(t) for t in ts]

    def get_dimension(self):
        return 300


def test_reference_pipeline_runs_on_this_packages_classes(reference, monkeypatch):
    from tokenizers import Tokenizer

    from verbatim_rag_amd.extractors import GpuModelSpanExtractor
    from verbatim_rag_amd.index import HotPathIndex
    from verbatim_rag_amd.pipeline import StaticVerbatimPipeline

    VerbatimIndex, VerbatimRAG = reference
    monkeypatch.setattr(vs._lib, "load", lambda: None)
    monkeypatch.setattr(vs._lib, "require_gpu", lambda: None)
    monkeypatch.setattr(vs, "DenseShard", _Dense)
    monkeypatch.setattr(vs, "SparseShard", _Sparse)
    docs = [f"The tower number {i} is tall. It stands in city {i % 4}. Visitors climb {i + 3} stairs! Was it built in {1800 + i}?" for i in range(30)]
    store = vs.GpuVectorStore(dense_dim=64, sparse_vocab=300)
    dense, sparse = Dense(), Sparse()
    ours = HotPathIndex(store, dense_provider=dense, sparse_provider=sparse)
    ours.add_chunks([f"c{i}" for i in range(30)], docs, metadatas=[{"document_id": f"d{i % 3}", "title": f"Doc {i}", "source": f"s{i}.md"} for i in range(30)])
    ext = GpuModelSpanExtractor(engine=RecordingEngine(), tokenizer=Tokenizer.from_file(os.path.join(G, "tokenizer.json")), threshold=0.5)

    theirs = VerbatimIndex(vector_store=store, dense_provider=dense, sparse_provider=sparse)
    for kw in (dict(text="Where is the tall tower?", k=5), dict(text="Visitors climb stairs", k=3, search_type="sparse"),
               dict(text="city", k=4, search_type="dense", filter='metadata["document_id"] == "d1"'),
               dict(text="Where is it?", k=5, hybrid_weights={"dense": 0.7, "sparse": 0.3}, rrf_k=20)):
        a, b = theirs.query(**kw), ours.query(**kw)
        assert [(x.id, x.score, x.text, x.metadata) for x in a] == [(x.id, x.score, x.text, x.metadata) for x in b] and 0 < len(a) <= kw["k"]

    rag = VerbatimRAG(index=theirs, k=5, extractor=ext, template_mode="static", llm_client=types.SimpleNamespace())
    pipe = StaticVerbatimPipeline(ours, ext, k=5)
    for question in ("Where is the tall tower?", "How many stairs do visitors climb?"):
        resp = rag.query(question)
        mine = pipe.query(question)
        assert resp.model_dump() == mine.model_dump()
        assert len(resp.documents) == 5 and any(d.highlights for d in resp.documents)
        for d in resp.documents:
            assert all(d.content[h.start:h.end] == h.text for h in d.highlights)
    assert [r.model_dump() for r in pipe.query_batch(["Where is the tall tower?", "How many stairs do visitors climb?"])] == \
        [rag.query(q).model_dump() for q in ("Where is the tall tower?", "How many stairs do visitors climb?")]
    assert "verbatim_rag.index" in sys.modules


def test_reference_async_transform_and_reranker_seams(reference, monkeypatch):
    """`VerbatimRAG.query_async` (-> `extract_spans_async`, extractors.py:48-54), `VerbatimTransform.transform` (dict
    contexts, transform.py:86-112) and the `reranker=` hook (core.py:125-140) with this package's classes plugged in."""
    import asyncio

    from tokenizers import Tokenizer
    from verbatim_core.transform import VerbatimTransform

    from verbatim_rag_amd.extractors import CoalescingSpanExtractor, GpuModelSpanExtractor
    from verbatim_rag_amd.index import HotPathIndex
    from verbatim_rag_amd.pipeline import StaticVerbatimPipeline
    from verbatim_rag_amd.rerankers import GpuCrossEncoderReranker

    VerbatimIndex, VerbatimRAG = reference
    monkeypatch.setattr(vs._lib, "load", lambda: None)
    monkeypatch.setattr(vs._lib, "require_gpu", lambda: None)
    monkeypatch.setattr(vs, "DenseShard", _Dense)
    monkeypatch.setattr(vs, "SparseShard", _Sparse)
    docs = [f"The tower number {i} is tall. It stands in city {i % 4}. Visitors climb {i + 3} stairs! Was it built in {1800 + i}?" for i in range(20)]
    store = vs.GpuVectorStore(dense_dim=64, sparse_vocab=300)
    ours = HotPathIndex(store, dense_provider=Dense(), sparse_provider=Sparse())
    ours.add_chunks([f"c{i}" for i in range(20)], docs, metadatas=[{"title": f"Doc {i}", "source": f"s{i}.md"} for i in range(20)])
    tok = Tokenizer.from_file(os.path.join(G, "tokenizer.json"))
    ext = GpuModelSpanExtractor(engine=RecordingEngine(), tokenizer=tok, threshold=0.5)
    theirs = VerbatimIndex(vector_store=store, dense_provider=ours.dense_provider, sparse_provider=ours.sparse_provider)
    question = "Where is the tall tower?"

    rag = VerbatimRAG(index=theirs, k=4, extractor=ext, template_mode="static", llm_client=types.SimpleNamespace())
    sync = rag.query(question).model_dump()
    assert asyncio.run(rag.query_async(question)).model_dump() == sync

    co = CoalescingSpanExtractor(ext, max_wait_ms=1.0)          # the per-query interface in front of the batch path
    try:
        rag_co = VerbatimRAG(index=theirs, k=4, extractor=co, template_mode="static", llm_client=types.SimpleNamespace())
        assert asyncio.run(rag_co.query_async(question)).model_dump() == sync
    finally:
        co.close()

    tr = VerbatimTransform(llm_client=types.SimpleNamespace(), extractor=ext, template_mode="static")
    out = tr.transform(question, [{"content": docs[0], "title": "T0", "source": "s0"}, {"text": docs[1]}])
    assert [d.content for d in out.documents] == docs[:2] and out.documents[0].title == "T0"
    assert all(d.content[h.start:h.end] == h.text for d in out.documents for h in d.highlights)

    # RAG-agnostic adapters (providers.py:40-84): SearchResult -> context dicts -> transform
    from verbatim_rag.providers import IndexProvider

    ctx = IndexProvider(theirs).retrieve(question, k=3)
    assert len(ctx) == 3 and all(c["content"] in docs and c["title"].startswith("Doc ") for c in ctx)
    assert [d.content for d in tr.transform(question, ctx).documents] == [c["content"] for c in ctx]
    assert asyncio.run(IndexProvider(theirs).retrieve_async(question, k=3)) == ctx

    class PairEngine:                                           # cross-encoder stand-in: score = overlap with the question
        max_seqs, max_tokens, max_seq_len, pair_labels = 64, 8192, 512, 1
        shape = types.SimpleNamespace(cls_token_id=1, sep_token_id=2)

        def pair_logits(self, seqs, types_):
            return np.asarray([[float(sum(t) % 17)] for t in seqs], np.float32)

    rr = GpuCrossEncoderReranker(PairEngine(), tok, rerank_k=3)
    with_rr = VerbatimRAG(index=theirs, k=4, extractor=ext, template_mode="static", llm_client=types.SimpleNamespace(), reranker=rr)
    mine = StaticVerbatimPipeline(ours, ext, k=4, reranker=rr)
    a, b = with_rr.query(question).model_dump(), mine.query(question).model_dump()
    assert a == b and [d["content"] for d in a["documents"]] != [d["content"] for d in sync["documents"]]


def test_reference_streaming_over_this_packages_classes(reference, monkeypatch):
    """`StreamingRAG.stream_query` (streaming.py:24-177): documents, then highlights (extract_spans in a worker thread),
    then the answer -- reads `doc.metadata.get(...)` and the extractor's dict directly."""
    import asyncio

    from tokenizers import Tokenizer
    from verbatim_rag.streaming import StreamingRAG

    from verbatim_rag_amd.extractors import GpuModelSpanExtractor
    from verbatim_rag_amd.index import HotPathIndex

    VerbatimIndex, VerbatimRAG = reference
    monkeypatch.setattr(vs._lib, "load", lambda: None)
    monkeypatch.setattr(vs._lib, "require_gpu", lambda: None)
    monkeypatch.setattr(vs, "DenseShard", _Dense)
    monkeypatch.setattr(vs, "SparseShard", _Sparse)
    docs = [f"The tower number {i} is tall. It stands in city {i % 4}. Visitors climb {i + 3} stairs!" for i in range(12)]
    store = vs.GpuVectorStore(dense_dim=64, sparse_vocab=300)
    HotPathIndex(store, dense_provider=Dense(), sparse_provider=Sparse()).add_chunks(
        [f"c{i}" for i in range(12)], docs, metadatas=[{"title": f"Doc {i}", "source": f"s{i}.md"} for i in range(12)])
    ext = GpuModelSpanExtractor(engine=RecordingEngine(), tokenizer=Tokenizer.from_file(os.path.join(G, "tokenizer.json")), threshold=0.5)
    rag = VerbatimRAG(index=VerbatimIndex(vector_store=store, dense_provider=Dense(), sparse_provider=Sparse()), k=3, extractor=ext,
                      template_mode="static", llm_client=types.SimpleNamespace())

    async def collect():
        return [ev async for ev in StreamingRAG(rag).stream_query("Where is the tall tower?")]

    events = asyncio.run(collect())
    kinds = [e.get("type") for e in events]
    assert "error" not in kinds and kinds[0] == "documents" and "highlights" in kinds and kinds[-1] == "answer", kinds
    final = rag.query("Where is the tall tower?")
    assert events[-1]["done"] is True and events[-1]["data"] == final.model_dump()
    docs_ev = next(e for e in events if e["type"] == "documents")["data"]
    assert [d["content"] for d in docs_ev] == [d.content for d in final.documents] and all(d["highlights"] == [] for d in docs_ev)
    hl_ev = next(e for e in events if e["type"] == "highlights")["data"]
    assert [[h["text"] for h in d["highlights"]] for d in hl_ev] == [[h.text for h in d.highlights] for d in final.documents]


def test_reference_ingest_path_fills_this_packages_store(reference, monkeypatch, tmp_path):
    """`VerbatimIndex.add_documents` (index.py:318-411: the reference's chunker, `_prepare_chunk_metadata`,
    `_generate_embeddings`, `_store_chunks` -> `store.add_vectors(ids=, dense_vectors=, sparse_vectors=, texts=,
    enhanced_texts=, metadatas=)`), then queries, a `document_id` filter built the reference's way, save / load."""
    from verbatim_rag.schema import DocumentSchema

    VerbatimIndex, _VerbatimRAG = reference
    monkeypatch.setattr(vs._lib, "load", lambda: None)
    monkeypatch.setattr(vs._lib, "require_gpu", lambda: None)
    monkeypatch.setattr(vs, "DenseShard", _Dense)
    monkeypatch.setattr(vs, "SparseShard", _Sparse)
    store = vs.GpuVectorStore(dense_dim=64, sparse_vocab=300)
    idx = VerbatimIndex(vector_store=store, dense_provider=Dense(), sparse_provider=Sparse())
    docs = [DocumentSchema(content=f"# Title {i}\n\nThe tower number {i} is tall. It stands in city {i}.\n\n## Section\n\nVisitors climb {i + 3} stairs!",
                           title=f"Doc {i}", source=f"s{i}.md") for i in range(6)]
    idx.add_documents(docs)
    assert len(store._ids) >= 6 and all(isinstance(v, (str, int, float, bool, type(None), list, dict)) for md in store._meta for v in md.values())
    hits = idx.query(text="The tower is tall", k=4)
    assert len(hits) == 4 and all(h.metadata["title"].startswith("Doc ") and h.text for h in hits)
    doc_id = hits[0].metadata["document_id"]
    only = idx.query(text="The tower is tall", k=10, filter=f'metadata["document_id"] == "{doc_id}"')
    assert only and all(h.metadata["document_id"] == doc_id for h in only)
    # accessors the reference builds on filter-only / vector-less queries (index.py:657-760) and the document records
    assert len(idx.get_all_chunks(limit=5)) == 5
    listed = idx.get_all_documents(limit=3)
    assert len(listed) == 3 and all(d["title"].startswith("Doc ") and d["content_type"] for d in listed)
    by_doc = idx.get_chunks_by_document(doc_id)
    assert by_doc and all(h.metadata["document_id"] == doc_id for h in by_doc)
    assert idx.inspect()["total_documents"] == 6
    rec = idx.get_document(doc_id)
    assert rec is not None and rec["id"] == doc_id and rec["title"].startswith("Doc ") and idx.get_document("nope") is None
    store.save(str(tmp_path / "idx"))
    again = VerbatimIndex(vector_store=vs.GpuVectorStore.load(str(tmp_path / "idx")), dense_provider=Dense(), sparse_provider=Sparse())
    assert [(h.id, h.score) for h in again.query(text="The tower is tall", k=4)] == [(h.id, h.score) for h in hits]
    assert again.get_document(doc_id) == rec
