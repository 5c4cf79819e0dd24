"""Retrieval parity at the single-GPU sizes BASELINE.json names, checked against the exact CPU oracle
(oracle/topk_ref.c, blocked OpenMP form -- the same fmaf chains as the scalar form, tests/test_oracle_golden.py):

  * configs[2]: a 10^6-document / ~1.28 * 10^8-nnz SELL-64 shard, 1 000 queries, k = 5 -- the 8-queries-per-pass kernel
    and the single-query kernel, ids AND scores equal to the oracle;
  * configs[3] (one GPU's slice of the 10^7-row index): 1.25 * 10^6 x 768 rows, fp32 (bit-exact on arbitrary data) and
    bf16 (bit-exact on bf16-representable grid data), 1 / 32 / 256 (bf16: / 4 096) queries;
  * both through the PUBLIC ingest path -- `GpuVectorStore.add_vectors` (milvus_base.py:90-127) -> `query_batch` /
    `query` (milvus_base.py:189-313) on a 10^6-chunk hybrid store: dense, sparse, hybrid (RRF), filtered, after a
    delete and after an append (dense append + sparse tail segment);
  * configs[4] slice: `StaticVerbatimPipeline.query_batch` over a hybrid store with the ModernBERT-large extractor at
    full depth -- spans, highlights and citation offsets equal to the same pipeline with the extractor's logits
    replaced by the fp32 oracle's (oracle/modernbert_np.py).
"""
import gc
import json
import os
import time
import types

import numpy as np
import pytest

from oracle import topk_ref as T
from tests import synth_corpus as S

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
VOCAB = 30522
SCALE = float(os.environ.get("VRAG_TEST_SCALE", "1"))            # < 1: host-logic dry runs on CPU stand-ins
N_SPARSE = int(1_000_000 * SCALE)
N_DENSE = int(1_250_000 * SCALE)
DIM = 768


def _note(key, value):
    """Timings for DESIGN.md: merged into gpurun_out/full_size_timings.json when that scratch directory exists."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if not os.path.isdir(root):
        return
    path = os.path.join(root, "full_size_timings.json")
    try:
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[key] = value
        json.dump(data, open(path, "w"), indent=1)
    except Exception:
        pass


@pytest.fixture(scope="module")
def sparse_rows():
    return S.sparse_corpus(N_SPARSE, VOCAB, seed=21)


def test_configs2_sparse_shard_1m_docs_1000_queries(sparse_rows):
    from verbatim_rag_amd.vector_stores import SparseShard

    ip, ix, vv = sparse_rows
    t0 = time.perf_counter()
    sh = SparseShard(VOCAB, ip, ix, vv)
    _note("sparse_shard_build_s", time.perf_counter() - t0)
    st = sh.stats()
    assert st["n_docs"] == N_SPARSE and st["nnz"] == len(ix) > 120 * N_SPARSE and st["padded_nnz"] < 1.02 * st["nnz"]
    _dq, (qp, qi, qv) = S.sparse_queries(1000, VOCAB, seed=22)
    k = 5
    t0 = time.perf_counter()
    s, i = sh.search_csr(qp, qi, qv, k)                        # 8 queries per pass
    _note("sparse_1000_queries_search_s", time.perf_counter() - t0)
    t0 = time.perf_counter()
    rs, ri = T.sparse_topk(ip, ix, vv, VOCAB, qp, qi, qv, k, blocked=True)
    _note("sparse_1000_queries_oracle_s", time.perf_counter() - t0)
    assert (ri >= 0).all()
    assert np.array_equal(i, ri)
    assert np.array_equal(s, rs)
    for q in range(0, 1000, 37):                               # the single-query kernel on a sample
        a, b = int(qp[q]), int(qp[q + 1])
        s1, i1 = sh.search_csr(np.asarray([0, b - a], np.int64), qi[a:b], qv[a:b], k)
        assert np.array_equal(i1[0], ri[q]) and np.array_equal(s1[0], rs[q]), q
    s64, i64 = sh.search_csr(qp[:5], qi[: qp[4]], qv[: qp[4]], 64)     # a full device page per query
    rs64, ri64 = T.sparse_topk(ip, ix, vv, VOCAB, qp[:5], qi[: qp[4]], qv[: qp[4]], 64, blocked=True)
    assert np.array_equal(i64, ri64) and np.array_equal(s64, rs64)
    sh.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_configs3_dense_shard_1_25m_rows(dtype):
    """fp32 rows: arbitrary values, the kernel's per-row fmaf chain equals the oracle's bit for bit.  bf16 rows: values on
    a bf16-exact grid (every product and sum exact), so the matrix-core path must agree exactly too, ties included."""
    from verbatim_rag_amd.vector_stores import DenseShard

    gen = S.dense_rows if dtype == "f32" else S.dense_grid
    X = gen(N_DENSE, DIM, seed=31)
    # bf16 rows: up to 4 096 queries -- batches of >= 64 take the tiled search (the shard read once per batch: rows x queries
    # on the encoder's GEMM kernel with the candidate-filter epilogue, csrc/topk.hip "dense, tiled batched search")
    nqs = (1, 32, 256) if dtype == "f32" else (1, 32, 256, 4096)
    Q = gen(nqs[-1], DIM, seed=32)
    sh = DenseShard(DIM, N_DENSE, dtype)
    t0 = time.perf_counter()
    sh.add(X)
    _note(f"dense_{dtype}_upload_s", time.perf_counter() - t0)
    k = 10
    t0 = time.perf_counter()
    rs, ri = T.dense_topk(X, Q, k, blocked=True)
    _note(f"dense_{dtype}_{nqs[-1]}_queries_oracle_s", time.perf_counter() - t0)
    try:
        for nq in nqs:
            t0 = time.perf_counter()
            s, i = sh.search(Q[:nq], k)
            # the FIRST call of this shape on the index (scratch sized and every route run once at ingest, csrc/topk.hip
            # dense_warm_query_path; round 5 paid 58 ms here for 256 fp32 queries) and a repeat, which is what profiles/README quotes
            _note(f"dense_{dtype}_{nq}_queries_first_call_s", time.perf_counter() - t0)
            assert np.array_equal(i, ri[:nq]), (dtype, nq)
            assert np.array_equal(s, rs[:nq]), (dtype, nq)
            dt = float("inf")
            for _rep in range(3):                  # the best of three: one repeat took 71 ms on a box once (a host hiccup -- the next run
                t0 = time.perf_counter()           # of the same build was back at 1.1 ms), a per-search allocation would slow all three
                s2, i2 = sh.search(Q[:nq], k)
                dt = min(dt, time.perf_counter() - t0)   # the search alone: the note below is file I/O, the comparisons host work
                assert np.array_equal(i2, i) and np.array_equal(s2, s)
            _note(f"dense_{dtype}_{nq}_queries_search_s", dt)
            if nq <= 256:
                assert dt < 0.02, "a repeated <= 256-query search takes milliseconds"
    finally:
        sh.close()
        del X
        gc.collect()


def _rrf_expect(rows_d, rows_s, top_k, rrf_k=60):
    from verbatim_rag_amd.vector_stores import rrf_merge_rows

    return rrf_merge_rows({"dense": rows_d, "sparse": rows_s}, top_k, {"dense": 0.5, "sparse": 0.5}, rrf_k)


def test_store_ingests_1m_chunks_and_answers_like_the_oracle(sparse_rows):
    from verbatim_rag_amd.vector_stores import GpuVectorStore

    ip, ix, vv = sparse_rows
    n = N_SPARSE
    X = S.dense_rows(n, DIM, seed=41)
    ids = [f"c{i}" for i in range(n)]
    texts = [f"chunk {i}" for i in range(n)]
    metas = [{"document_id": f"d{i >> 6}", "n": i} for i in range(n)]
    st = GpuVectorStore(dense_dim=DIM, sparse_vocab=VOCAB)
    t0 = time.perf_counter()
    st.add_vectors(ids, X, (ip, ix, vv), texts, texts, metas)
    t_ingest = time.perf_counter() - t0
    _note("store_add_vectors_1m_s", t_ingest)
    assert t_ingest < 60, f"add_vectors took {t_ingest:.1f} s for 10^6 chunks"
    del X
    gc.collect()
    nq, k = 256, 5
    Q = S.dense_rows(nq, DIM, seed=42)
    dq, (qp, qi, qv) = S.sparse_queries(nq, VOCAB, seed=43)
    t0 = time.perf_counter()
    got_d = st.query_batch(dense_queries=Q, search_type="dense", top_k=k)            # includes the flush (upload, SELL build)
    _note("store_first_query_batch_with_flush_s", time.perf_counter() - t0)
    rows = st._dense_rows.data                                                       # the unit rows the index holds
    uq = st._unit_queries(Q)
    rs_d, ri_d = T.dense_topk(rows, uq, 2 * k, blocked=True)
    rs_s, ri_s = T.sparse_topk(ip, ix, vv, VOCAB, qp, qi, qv, 2 * k, blocked=True)

    def check(got, want_rows, want_scores, exact_scores=True):
        assert len(got) == len(want_rows)
        for q, rs in enumerate(got):
            want = [int(r) for r in want_rows[q] if r >= 0]
            assert [r.id for r in rs] == [f"c{r}" for r in want], q
            assert [r.text for r in rs] == [f"chunk {r}" for r in want]
            assert [r.metadata for r in rs] == [{"document_id": f"d{r >> 6}", "n": r} for r in want]
            if exact_scores:
                assert [r.score for r in rs] == [float(v) for v, r in zip(want_scores[q], want_rows[q]) if r >= 0], q

    check(got_d, ri_d[:, :k], rs_d[:, :k])
    check(st.query_batch(sparse_queries=dq, search_type="sparse", top_k=k), ri_s[:, :k], rs_s[:, :k])
    hyb_rows, hyb_dist = _rrf_expect(ri_d, ri_s, k)
    check(st.query_batch(dense_queries=Q, sparse_queries=dq, search_type="hybrid", top_k=k), hyb_rows, hyb_dist)
    for q in (0, 100, 255):                                                          # the per-query entry point
        one = st.query(dense_query=Q[q].tolist(), sparse_query=dq[q], top_k=k, search_type="hybrid")
        check([one], hyb_rows[q:q + 1], hyb_dist[q:q + 1])
    # a per-document filter: 64 of 10^6 rows pass -> the masked subset shard answers
    flt = 'metadata["document_id"] == "d77"'
    lo = 77 * 64
    sub_s, sub_i = T.dense_topk(rows[lo:lo + 64], uq[:8], k)
    check(st.query_batch(dense_queries=Q[:8], search_type="dense", top_k=k, filter=flt), sub_i + lo, sub_s)
    # deletes act before the search
    gone = [f"c{int(r)}" for r in ri_d[:4, 0]]
    st.delete(gone)
    after = st.query_batch(dense_queries=Q[:4], search_type="dense", top_k=k)
    for q in range(4):
        want = [int(r) for r in ri_d[q] if f"c{int(r)}" not in gone][:k]
        assert [r.id for r in after[q]] == [f"c{r}" for r in want]
    # an append: dense rows join the resident shard, sparse rows form a tail segment beside the main SELL image
    m = 1000
    X2 = S.dense_rows(m, DIM, seed=44)
    ip2, ix2, vv2 = S.sparse_corpus(m, VOCAB, seed=45)
    X2[:nq] = Q * np.float32(3.0)                                                   # row n + q is query q's best dense hit
    st.add_vectors([f"c{n + i}" for i in range(m)], X2, (ip2, ix2, vv2), [f"chunk {n + i}" for i in range(m)],
                   [f"chunk {n + i}" for i in range(m)], [{"document_id": f"d{(n + i) >> 6}", "n": n + i} for i in range(m)])
    top = st.query_batch(dense_queries=Q, search_type="dense", top_k=1)
    assert [r[0].id for r in top] == [f"c{n + q}" for q in range(nq)]
    assert len(st._sparse_parts) == 2 and st._sparse_parts[1][1:] == (n, m)
    all_ip = np.concatenate([ip, ip2[1:] + ip[-1]])
    rs_s2, ri_s2 = T.sparse_topk(all_ip, np.concatenate([ix, ix2]), np.concatenate([vv, vv2]), VOCAB, qp[:17], qi[: qp[16]], qv[: qp[16]],
                                 k + len(gone), blocked=True)
    alive = st._alive.data
    order = np.argsort(~alive[ri_s2], axis=1, kind="stable")[:, :k]                   # deleted rows out, ranking kept
    check(st.query_batch(sparse_queries=dq[:16], search_type="sparse", top_k=k), np.take_along_axis(ri_s2, order, axis=1),
          np.take_along_axis(rs_s2, order, axis=1))
    for shard, _b, _n in st._sparse_parts:
        shard.close()
    st._dense.close()


def test_configs4_slice_pipeline_with_the_large_extractor_equals_the_oracle_pipeline():
    """BASELINE configs[4] on one GPU, small corpus: hybrid (SPLADE + dense) retrieval -> top-5 -> ModernBERT-large
    sentence classifier (28 layers at full width) -> static template -> citations.  The reference pipeline's arithmetic
    is `QAModel.forward` in fp32 (extractor_models/model.py:59-117); the second run replaces the extractor's logits
    with the fp32 oracle's (oracle/modernbert_np.py) behind the same host code: spans, highlights and citation offsets
    must be identical."""
    from tokenizers import Tokenizer

    from oracle import modernbert_np as O
    from verbatim_rag_amd.embedding_providers import GpuDenseProvider, GpuSpladeProvider
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape
    from verbatim_rag_amd.extractors import GpuModelSpanExtractor, SpanExtractor, select_sentences
    from verbatim_rag_amd.index import HotPathIndex
    from verbatim_rag_amd.pipeline import StaticVerbatimPipeline
    from verbatim_rag_amd.vector_stores import GpuVectorStore
    from verbatim_rag_amd.weights import random_init, random_qa_head

    tok = Tokenizer.from_file(os.path.join(G, "tokenizer.json"))
    with open(os.path.join(G, "host_fixtures.json")) as f:
        fx = json.load(f)
    # corpus: sentences of the captured config-1 documents re-dealt into 400 chunks of 3 - 7 sentences
    sents = [s for d in fx["config1"]["docs"] for s in d.replace("?", ".").split(". ") if len(s) > 12]
    rng = np.random.default_rng(9)
    chunks = []
    for i in range(400):
        pick = rng.choice(len(sents), size=int(rng.integers(3, 8)), replace=False)
        chunks.append(" ".join(sents[j].rstrip(".") + (" number %d." % i if n == 0 else ".") for n, j in enumerate(pick)))
    tiny = dict(vocab_size=512, hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=192,
                pad_token_id=0, cls_token_id=1, sep_token_id=2)
    z = np.load(os.path.join(G, "encoder_tiny.npz"))
    emb = EncoderEngine(ModernBertShape(**tiny), O.random_weights(O.EncoderConfig(**tiny), seed=7), max_tokens=16384, max_seqs=256,
                        max_seq_len=512, max_ranges=256)
    emb.set_mlm_head(z["mlm_head.dense.weight"], z["mlm_head.norm.weight"], z["mlm_decoder.bias"])
    store = GpuVectorStore(dense_dim=128, sparse_vocab=512)
    index = HotPathIndex(store, dense_provider=GpuDenseProvider(emb, tok, pooling="mean"), sparse_provider=GpuSpladeProvider(emb, tok))
    index.add_chunks([f"k{i}" for i in range(len(chunks))], chunks,
                     metadatas=[{"title": f"Doc {i}", "source": f"s{i}.md", "document_id": f"d{i}"} for i in range(len(chunks))])
    base = ModernBertShape.large()
    shape = ModernBertShape(**{**base.__dict__, "vocab_size": 512, "pad_token_id": 0, "cls_token_id": 1, "sep_token_id": 2})
    w = random_init(shape, seed=404)
    qa_w, qa_b = random_qa_head(shape)
    eng = EncoderEngine(shape, w, max_tokens=16384, max_seqs=64, max_seq_len=512, max_ranges=1024)
    eng.set_qa_head(qa_w, qa_b)
    gpu_ext = GpuModelSpanExtractor(engine=eng, tokenizer=tok, threshold=0.5)
    cfg = O.EncoderConfig(vocab_size=512, hidden_size=shape.hidden_size, num_hidden_layers=shape.num_hidden_layers,
                          num_attention_heads=shape.num_attention_heads, intermediate_size=shape.intermediate_size,
                          global_attn_every_n_layers=shape.global_attn_every_n_layers, local_attention=shape.local_attention,
                          global_rope_theta=shape.global_rope_theta, local_rope_theta=shape.local_rope_theta,
                          norm_eps=shape.norm_eps, pad_token_id=0, cls_token_id=1, sep_token_id=2)
    cache = {}

    def oracle_logits(question, text, smp):
        key = (question, text)
        if key not in cache:
            hid = O.encoder_forward(cfg, w, np.asarray(smp.input_ids, np.int32))
            cache[key] = O.qa_sentence_logits(hid, smp.sentence_boundaries, qa_w, qa_b)
        return cache[key]

    class OracleExtractor(SpanExtractor):
        """The extractor's host code (split, packing, threshold select) around the fp32 CPU arithmetic."""

        def extract_spans(self, question, search_results):
            out = {}
            texts = [getattr(r, "text", "") for r in search_results]
            all_sents, samples = gpu_ext.pack_qa(question, texts)
            for text, raw, smp in zip(texts, all_sents, samples):
                out[text] = [] if smp is None else select_sentences(oracle_logits(question, text, smp), raw, gpu_ext.threshold)
            return out

    questions = ["Where is the tower?", "Who built the old bridge?", "When was the museum opened?"]
    oracle_pipe = StaticVerbatimPipeline(index, OracleExtractor(), k=5)
    for q in questions:                                     # first pass: the oracle's sentence probabilities for every retrieved pair
        oracle_pipe.query(q)
    probs = np.sort(np.concatenate([O.softmax_rows(lg)[:, 1] for lg in cache.values()]))
    # the decision threshold sits in the middle of the widest gap between two probabilities in [0.3, 0.7]: the comparison
    # below is about arithmetic within 1e-3, not about a sentence that happens to sit on the threshold
    mid = probs[(probs > 0.3) & (probs < 0.7)]
    assert len(mid) >= 2
    g = int(np.argmax(np.diff(mid)))
    gpu_ext.threshold = float((mid[g] + mid[g + 1]) / 2)
    assert mid[g + 1] - mid[g] > 4e-3
    got = StaticVerbatimPipeline(index, gpu_ext, k=5).query_batch(questions)
    want = [oracle_pipe.query(q) for q in questions]
    eng.close()
    emb.close()
    n_cited = 0
    for g, wnt in zip(got, want):
        g, wnt = g.model_dump(), wnt.model_dump()
        assert [d["highlights"] for d in g["documents"]] == [d["highlights"] for d in wnt["documents"]]    # citation offsets
        assert g["structured_answer"]["citations"] == wnt["structured_answer"]["citations"]
        assert g["answer"] == wnt["answer"]
        n_cited += len(g["structured_answer"]["citations"])
    assert n_cited > 0


class _LookupProvider:
    """Query-side stand-in for an embedding provider in the composed run below: question text -> a prepared vector (the
    encoders behind the real providers are oracle-tested on their own: test_heads_gpu.py, test_bert_gpu.py,
    test_splade_real_vocab_gpu.py; what is composed here is store + extractor + pipeline at configs[4]'s per-GPU size)."""

    def __init__(self, table, dim):
        self.table, self.dim = table, dim

    def embed_text(self, text):
        return self.table[text]

    def embed_queries(self, texts):
        return [self.table[t] for t in texts]

    def embed_batch(self, texts):
        return [self.table[t] for t in texts]

    def get_dimension(self):
        return self.dim


def test_configs4_one_gpu_slice_composed_1_25m_hybrid_rows_large_extractor_1024_queries():
    """BASELINE configs[4] as ONE GPU of the 8 sees it, composed: a 1.25 * 10^6-row hybrid store
    (synthetic dense rows + SPLADE-shaped CSR rows, real multi-sentence chunk texts), 1 024 concurrent questions through
    `StaticVerbatimPipeline.query_batch` -- batched hybrid retrieval (tiled dense search + batched sparse search + RRF),
    top-5, the ModernBERT-large sentence classifier at full depth over the 5 120 (question, chunk) pairs, static template,
    citations.  On a 16-question sample the responses (answers, highlights, citation offsets) must equal the same pipeline
    with the extractor's logits replaced by the fp32 oracle's.  Wall times go to gpurun_out/full_size_timings.json."""
    from tokenizers import Tokenizer

    from oracle import modernbert_np as O
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape
    from verbatim_rag_amd.extractors import GpuModelSpanExtractor, SpanExtractor, select_sentences
    from verbatim_rag_amd.index import HotPathIndex
    from verbatim_rag_amd.pipeline import StaticVerbatimPipeline
    from verbatim_rag_amd.vector_stores import GpuVectorStore
    from verbatim_rag_amd.weights import random_init, random_qa_head

    n, nq, n_sample = N_DENSE, 1024, int(os.environ.get("VRAG_TEST_ORACLE_SAMPLE", "16"))   # 7 s of CPU oracle per question
    tok = Tokenizer.from_file(os.path.join(G, "tokenizer.json"))
    with open(os.path.join(G, "host_fixtures.json")) as f:
        fx = json.load(f)
    sents = [s for d in fx["config1"]["docs"] for s in d.replace("?", ".").split(". ") if len(s) > 12]
    rng = np.random.default_rng(19)
    pool = []
    for i in range(512):                                   # 512 distinct chunk texts of 3 - 7 sentences, dealt over the rows
        pick = rng.choice(len(sents), size=int(rng.integers(3, 8)), replace=False)
        pool.append(" ".join(sents[j].rstrip(".") + (" number %d." % i if m == 0 else ".") for m, j in enumerate(pick)))
    texts = [pool[(i * 2654435761) % 512] for i in range(n)]
    t0 = time.perf_counter()
    X = S.dense_rows(n, DIM, seed=51)
    ip, ix, vv = S.sparse_corpus(n, VOCAB, seed=52)
    store = GpuVectorStore(dense_dim=DIM, sparse_vocab=VOCAB)
    store.add_vectors([f"c{i}" for i in range(n)], X, (ip, ix, vv), texts, texts,
                      [{"title": f"Doc {i >> 6}", "source": f"s{i >> 6}.md", "document_id": f"d{i >> 6}"} for i in range(n)])
    del X
    gc.collect()
    _note("composed_store_build_1_25m_s", time.perf_counter() - t0)
    Q = S.dense_rows(nq, DIM, seed=53)
    dq, _csr = S.sparse_queries(nq, VOCAB, seed=54)
    words = ["tower", "bridge", "museum", "river", "engineer", "harbour", "castle", "library"]
    questions = [f"Where is the {words[i % 8]} number {i}?" for i in range(nq)]
    index = HotPathIndex(store, dense_provider=_LookupProvider({q: Q[i].tolist() for i, q in enumerate(questions)}, DIM),
                         sparse_provider=_LookupProvider({q: dq[i] for i, q in enumerate(questions)}, VOCAB))
    base = ModernBertShape.large()
    shape = ModernBertShape(**{**base.__dict__, "vocab_size": 512, "pad_token_id": 0, "cls_token_id": 1, "sep_token_id": 2})
    w = random_init(shape, seed=404)
    qa_w, qa_b = random_qa_head(shape)
    eng = EncoderEngine(shape, w, max_tokens=262144, max_seqs=2048, max_seq_len=512, max_ranges=16384, micro_batch_tokens=65536)
    eng.set_qa_head(qa_w, qa_b)
    gpu_ext = GpuModelSpanExtractor(engine=eng, tokenizer=tok, threshold=0.5)
    cfg = O.EncoderConfig(vocab_size=512, hidden_size=shape.hidden_size, num_hidden_layers=shape.num_hidden_layers,
                          num_attention_heads=shape.num_attention_heads, intermediate_size=shape.intermediate_size,
                          global_attn_every_n_layers=shape.global_attn_every_n_layers, local_attention=shape.local_attention,
                          global_rope_theta=shape.global_rope_theta, local_rope_theta=shape.local_rope_theta,
                          norm_eps=shape.norm_eps, pad_token_id=0, cls_token_id=1, sep_token_id=2)
    cache = {}

    def oracle_logits(question, text, smp):
        key = (question, text)
        if key not in cache:
            hid = O.encoder_forward(cfg, w, np.asarray(smp.input_ids, np.int32))
            cache[key] = O.qa_sentence_logits(hid, smp.sentence_boundaries, qa_w, qa_b)
        return cache[key]

    class OracleExtractor(SpanExtractor):
        def extract_spans(self, question, search_results):
            docs = [getattr(r, "text", "") for r in search_results]
            all_sents, samples = gpu_ext.pack_qa(question, docs)
            out = {}
            for text, raw, smp in zip(docs, all_sents, samples):
                out[text] = [] if smp is None else select_sentences(oracle_logits(question, text, smp), raw, gpu_ext.threshold)
            return out

    try:
        pipe = StaticVerbatimPipeline(index, gpu_ext, k=5)
        pipe.query_batch(questions[:64])                       # warm-up: store flush (SELL build, uploads), chunk cache
        t0 = time.perf_counter()
        got = pipe.query_batch(questions)
        t_batch = time.perf_counter() - t0
        _note("composed_1024_queries_query_batch_s", t_batch)
        _note("composed_queries_per_s", nq / t_batch)
        assert len(got) == nq and all(1 <= len(r.documents) <= 5 for r in got)
        # the retrieval both arms below share, against the CPU oracle (it was checked elsewhere only): the hybrid
        # call the pipeline makes -- exact dense top-10 over the unit rows the index holds, exact sparse top-10, RRF -- for the sample
        sample = list(range(0, nq, max(1, nq // n_sample)))[:n_sample]
        qp_s, qi_s, qv_s = [0], [], []
        for i in sample:
            qi_s += list(dq[i].keys())
            qv_s += list(dq[i].values())
            qp_s.append(len(qi_s))
        rs_d, ri_d = T.dense_topk(store._dense_rows.data, store._unit_queries(Q[sample]), 10, blocked=True)
        rs_s, ri_s = T.sparse_topk(ip, ix, vv, VOCAB, np.asarray(qp_s, np.int64), np.asarray(qi_s, np.int32), np.asarray(qv_s, np.float32), 10,
                                   blocked=True)
        hyb_rows, hyb_dist = _rrf_expect(ri_d, ri_s, 5)
        hits = store.query_batch(dense_queries=Q[sample], sparse_queries=[dq[i] for i in sample], search_type="hybrid", top_k=5)
        for j in range(len(sample)):
            want = [int(r) for r in hyb_rows[j] if r >= 0]
            assert [h.id for h in hits[j]] == [f"c{r}" for r in want], sample[j]
            assert [h.score for h in hits[j]] == [float(v) for v, r in zip(hyb_dist[j], hyb_rows[j]) if r >= 0], sample[j]
            assert [h.text for h in hits[j]] == [texts[r] for r in want]
        del ip, ix, vv
        # oracle pass over the sample: probabilities first, then a threshold in the widest gap (the comparison is about
        # arithmetic within 1e-3, not about a sentence that happens to sit on the threshold)
        oracle_pipe = StaticVerbatimPipeline(index, OracleExtractor(), k=5)
        t0 = time.perf_counter()
        for i in sample:
            oracle_pipe.query(questions[i])
        _note("composed_oracle_16_queries_s", time.perf_counter() - t0)
        probs = np.sort(np.concatenate([O.softmax_rows(lg)[:, 1] for lg in cache.values()]))
        mid = probs[(probs > 0.3) & (probs < 0.7)]
        assert len(mid) >= 2
        g = int(np.argmax(np.diff(mid)))
        gpu_ext.threshold = float((mid[g] + mid[g + 1]) / 2)
        assert mid[g + 1] - mid[g] > 2e-3
        got_s = pipe.query_batch([questions[i] for i in sample])
        want = [oracle_pipe.query(questions[i]) for i in sample]
        n_cited = 0
        for a, b in zip(got_s, want):
            a, b = a.model_dump(), b.model_dump()
            assert [d["highlights"] for d in a["documents"]] == [d["highlights"] for d in b["documents"]]    # citation offsets
            assert a["structured_answer"]["citations"] == b["structured_answer"]["citations"]
            assert a["answer"] == b["answer"]
            n_cited += len(a["structured_answer"]["citations"])
        assert n_cited > 0
    finally:
        eng.close()
        for shard, _b, _n in store._sparse_parts:
            shard.close()
        store._dense.close()
