"""Host-side logic vs golden vectors captured from the imported reference
(tests/golden/gen_golden.py; runs on CPU, no GPU and no reference needed)."""
import json
import logging
import os
import types

import numpy as np
import pytest

import verbatim_rag_amd  # noqa: F401
from verbatim_rag_amd import packing
from verbatim_rag_amd.extractors import select_sentences, token_spans_to_char_spans
from verbatim_rag_amd.index import HotPathIndex
from verbatim_rag_amd.pipeline import StaticVerbatimPipeline
from verbatim_rag_amd.response_builder import ResponseBuilder
from verbatim_rag_amd.vector_stores import (SearchResult, VectorStore, convert_hits_to_results, merge_hybrid_results,
                                            normalize_weights, sanitize_hybrid_weights)

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(G, "host_fixtures.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def tok():
    from tokenizers import Tokenizer

    return packing.TokenizerAdapter(Tokenizer.from_file(os.path.join(G, "tokenizer.json")))


def test_sentence_split_matches_reference(fx):
    for case in fx["sentence_split"]:
        assert packing.split_into_sentences(case["text"]) == case["sentences"], case["text"]


def test_packer_matches_reference(fx, tok):
    assert tok.sep_token_id == 2 and tok.cls_token_id == 1
    n_drop = 0
    for case in fx["packer"]:
        budget = case["max_length"] - 2
        q = tok.ids(case["question"], True, budget)
        sents = tok.ids_batch(case["sentences"], budget)
        smp = packing.encode_question_and_sentences(q, sents, tok.sep_token_id, case["max_length"])
        assert smp.input_ids == case["input_ids"]
        assert [list(b) for b in smp.sentence_boundaries] == case["sentence_boundaries"]
        n_drop += len(smp.sentence_boundaries) < len(case["sentences"])
    assert n_drop > 5  # the overflow-drop branch is exercised


def test_packer_hf_fast_tokenizer_agrees(fx):
    transformers = pytest.importorskip("transformers")
    t = transformers.PreTrainedTokenizerFast(tokenizer_file=os.path.join(G, "tokenizer.json"), cls_token="[CLS]",
                                             sep_token="[SEP]", pad_token="[PAD]", unk_token="[UNK]")
    ad = packing.TokenizerAdapter(t)
    for case in fx["packer"][:12]:
        budget = case["max_length"] - 2
        smp = packing.encode_question_and_sentences(ad.ids(case["question"], True, budget),
                                                    ad.ids_batch(case["sentences"], budget), ad.sep_token_id,
                                                    case["max_length"])
        assert smp.input_ids == case["input_ids"]


def test_budget_warning_known_answer(caplog):
    # tests/test_extractors.py:29-83 of the reference: fake tokenizer, max_length 7
    q = [0, 1]                       # add_special_tokens=True -> 2 ids, last != sep
    sents = [[0, 1], [0, 1, 2]]      # "one two", "three four five"
    with caplog.at_level(logging.WARNING, logger="verbatim_rag_amd.packing"):
        smp = packing.encode_question_and_sentences(q, sents, sep_token_id=2, max_length=7)
    assert "exceeded the 7-token budget; dropping 1 sentence(s)" in caplog.text
    assert len(smp.sentence_boundaries) == 1


def test_valid_boundaries_clamp_and_skip():
    assert packing.valid_boundaries([(3, 20), (100, 400), (50, 10), (-1, 5)], 130) == [(3, 20), (100, 129)]


def test_threshold_selection_matches_reference(fx):
    e = fx["extract_e2e"]
    for run in e["runs"]:
        for text, logits in zip(run["texts"], e["logits"]):
            sents = packing.split_into_sentences(text)
            got = select_sentences(np.asarray(logits, np.float32).reshape(-1, 2), sents, run["threshold"])
            assert got == run["spans"][text]


def test_rrf_matches_reference(fx):
    def hits(ids):
        return [{"id": i, "distance": 1.0 - 0.01 * n, "entity": {"text": f"t{i}", "enhanced_text": f"e{i}",
                                                                "metadata": json.dumps({"document_id": f"d{i}"})}}
                for n, i in enumerate(ids)]

    for c in fx["rrf"]:
        rbm = {"dense": hits(c["dense"]), "sparse": hits(c["sparse"])}
        if c["full_text"] is not None:
            rbm["full_text"] = hits(c["full_text"])
        assert normalize_weights(rbm, c["weights"]) == c["normalized"]
        merged = merge_hybrid_results(rbm, c["top_k"], c["weights"], rrf_k=c["rrf_k"])
        assert [h["id"] for h in merged] == c["ids"]
        assert [h["distance"] for h in merged] == c["distances"]  # float64, bit-exact
        res = convert_hits_to_results(merged)
        assert [r.score for r in res] == c["result_scores"]
        assert [r.metadata for r in res] == c["result_metadata"]
    for c in fx["sanitize"]:
        assert sanitize_hybrid_weights(c["in"]) == c["out"]
    with pytest.raises(ValueError):
        sanitize_hybrid_weights({})


def test_highlight_offsets_match_reference(fx):
    rb = ResponseBuilder()
    for c in fx["highlights"]:
        got = [h.model_dump() for h in rb._create_highlights(c["text"], c["spans"])]
        assert got == c["highlights"]
    # reference KAT tests/test_response_builder.py:12-17
    h = rb._create_highlights("The cat sat on the mat.", ["cat"])
    assert (h[0].start, h[0].end) == (4, 7)


def test_build_response_matches_reference(fx):
    c = fx["build_response"]
    sr = [types.SimpleNamespace(text=t, metadata={"title": f"T{i}", "source": f"S{i}"}) for i, t in enumerate(c["texts"])]
    resp = ResponseBuilder().build_response("Q?", "An answer.", sr, c["relevant"], display_span_count=c["display_span_count"])
    assert resp.model_dump() == c["response"]


class _RecStore(VectorStore):
    enable_full_text = False

    def __init__(self):
        self.calls = []

    def add_vectors(self, *a, **k):
        self.calls.append(("add", a, k))

    def query(self, **kw):
        self.calls.append(kw)
        return []

    def delete(self, ids):
        pass


class _D:
    def embed_text(self, t): return [0.0, 1.0]
    def embed_batch(self, ts): return [[0.0, 1.0]] * len(ts)
    def get_dimension(self): return 2


class _S:
    def embed_text(self, t): return {1: 0.5}
    def embed_batch(self, ts): return [{1: 0.5}] * len(ts)
    def get_dimension(self): return 10


def test_index_query_dispatch_matches_reference(fx):
    for t in fx["index_query_trace"]:
        d = _D() if t["providers"] in ("both", "dense") else None
        s = _S() if t["providers"] in ("both", "sparse") else None
        store = _RecStore()
        idx = HotPathIndex(store, dense_provider=d, sparse_provider=s)
        if "error" in t:
            with pytest.raises(Exception):
                idx.query(**t["kwargs"])
            continue
        idx.query(**t["kwargs"])
        call = {k: (v if not isinstance(v, (list, dict)) or k in ("hybrid_weights", "search_params") else "<vec>")
                for k, v in store.calls[-1].items()}
        assert call == t["store_call"], t


def test_index_requires_a_provider():
    with pytest.raises(ValueError):
        HotPathIndex(_RecStore())


def test_ingest_batches_of_2000():
    store = _RecStore()
    idx = HotPathIndex(store, sparse_provider=_S())
    n = 4500
    idx.add_chunks([f"c{i}" for i in range(n)], ["t"] * n)
    assert [len(c[1][0]) for c in store.calls] == [2000, 2000, 500]


def test_config1_static_pipeline_matches_reference(fx):
    """BASELINE config 1 plumbing: retrieve k=5 -> extract -> static template -> highlights/citations,
    with the extractor's device step replaced by the golden logits (host logic only on CPU)."""
    c = fx["config1"]
    e = fx["extract_e2e"]
    logits_by_text = {t: lg for t, lg in zip(e["runs"][0]["texts"], e["logits"])}

    class CannedStore(_RecStore):
        def query(self, **kw):
            self.calls.append(kw)
            return [SearchResult(id=f"c{i}", score=1.0 - 0.1 * i, metadata={"title": f"Doc {i}", "source": f"src{i}.md"},
                                 text=c["docs"][i], enhanced_text=c["docs"][i]) for i in range(kw.get("top_k", 5))]

    class GoldenLogitExtractor:
        def extract_spans(self, question, results):
            return {r.text: select_sentences(np.asarray(logits_by_text[r.text], np.float32).reshape(-1, 2),
                                             packing.split_into_sentences(r.text), c["threshold"]) for r in results}

    store = CannedStore()
    pipe = StaticVerbatimPipeline(HotPathIndex(store, sparse_provider=_S()), GoldenLogitExtractor(), k=5)
    resp = pipe.query(c["question"])
    assert store.calls[-1]["search_type"] == "sparse" and store.calls[-1]["top_k"] == 5
    assert resp.model_dump() == c["response"]


def test_token_spans_to_char_spans():
    ctx = "Alpha beta gamma delta epsilon zeta eta theta iota kappa lambda mu."
    words, offs, pos = ctx.replace(".", "").split(), [], 0
    for w in words:
        a = ctx.index(w, pos)
        offs.append((a, a + len(w)))
        pos = a + len(w)
    probs = [0.9, 0.9, 0.1, 0.9, 0.9, 0.9, 0.9, 0.1, 0.1, 0.1, 0.1, 0.9]
    out = token_spans_to_char_spans(probs, offs, ctx, 0.5, min_span_chars=10, merge_gap_chars=8)
    assert out == ["Alpha beta gamma delta epsilon zeta eta"]            # gap "gamma" (7 chars) merged, "mu" dropped
    out = token_spans_to_char_spans(probs, offs, ctx, 0.5, min_span_chars=1, merge_gap_chars=0)
    assert out == ["Alpha beta", "delta epsilon zeta eta", "mu"]
    assert all(s in ctx for s in out)


def test_filter_expression_subset():
    import verbatim_rag_amd  # noqa: F401
    from verbatim_rag_amd.vector_stores import parse_filter

    md = [{"document_id": "d1", "year": 2020, "lang": "en"}, {"document_id": "d2", "year": 2021, "lang": "de"},
          {"document_id": "d3", "lang": "en"}]

    def sel(expr):
        p = parse_filter(expr)
        return [m["document_id"] for m in md if p(m)]

    assert sel('metadata["document_id"] == "d2"') == ["d2"]            # Local dialect (index.py:738)
    assert sel('document_id == "d2"') == ["d2"]                          # Cloud dialect (index.py:736)
    assert sel("metadata['lang'] != 'en'") == ["d2"]
    assert sel('document_id in ["d1", "d3"] and lang == "en"') == ["d1", "d3"]
    assert sel('year == 2021 or document_id == "d3"') == ["d2", "d3"]
    assert sel('not (lang == "en") || year == 2020') == ["d1", "d2"]
    assert sel('(year == 2020 || year == 2021) && lang == "de"') == ["d2"]
    assert sel("year > 2020") == ["d2"] and sel('metadata["year"] <= 2020 and lang >= "en"') == ["d1"]   # numbers / text order
    assert sel('year < "3000"') == [] and sel("lang < 5") == []          # a bound only matches metadata of its own kind
    for bad in ("a >", "a >> 3", "a > [1]", 'metadata["x"] like "y%"', 'document_id == ', 'document_id == "d1" extra', "", '== "d1"'):
        with pytest.raises(ValueError):
            parse_filter(bad)

