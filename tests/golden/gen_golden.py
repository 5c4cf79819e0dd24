#!/usr/bin/env python3
"""Generates the golden fixtures in tests/golden/ by IMPORTING the reference and `transformers`
in the build container (where /root/reference exists).  Nothing of the reference travels: only
inputs and expected outputs are written (json / npz).  Re-run: `python tests/golden/gen_golden.py`.

Recipe: stub `rapidfuzz` / `openai` (imported at module top by
verbatim_core/extractors.py:18, llm_client.py:15-18), shim `tokenizer.encode_plus` (removed in
transformers 5; dataset.py:131,161 call it) and call the unbound `QAModel.forward` on a namespace
holding `.bert` and `.classifier` (model.py:75,112 only touch those two attributes).
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def _install_stubs():
    d = tempfile.mkdtemp(prefix="vrag_stubs_")
    os.makedirs(os.path.join(d, "rapidfuzz"))
    open(os.path.join(d, "rapidfuzz", "__init__.py"), "w").write("")
    open(os.path.join(d, "rapidfuzz", "fuzz.py"), "w").write(
        "def partial_ratio_alignment(*a, **k):\n    raise RuntimeError('rapidfuzz stub')\n")
    os.makedirs(os.path.join(d, "openai"))
    open(os.path.join(d, "openai", "__init__.py"), "w").write(
        "class OpenAI:\n    def __init__(self,*a,**k): pass\nclass AsyncOpenAI(OpenAI):\n    pass\n")
    sys.path[:0] = [d, os.path.join(REF, "packages", "core"), REF]


# ----------------------------------------------------------------------------------------------
# synthetic corpus + tokenizer
# ----------------------------------------------------------------------------------------------
WORDS = ("the quick brown fox jumps over lazy dog tower paris iron built year tall meters visitors "
         "river city bridge stone engineer opened museum garden light night climb stairs lift wind "
         "steel design world fair paint color history france capital famous landmark ticket view "
         "north south east west floor summit restaurant glass cable radio antenna height record old new").split()


def make_corpus(rng, n_docs=10):
    docs = []
    for d in range(n_docs):
        sents = []
        for s in range(int(rng.integers(3, 7))):
            n = int(rng.integers(5, 14))
            ws = [WORDS[int(i)] for i in rng.integers(0, len(WORDS), size=n)]
            ws[0] = ws[0].capitalize()
            end = ".!?"[int(rng.integers(0, 3))]
            sents.append(" ".join(ws) + end)
        docs.append(" ".join(sents))
    return docs


def build_tokenizer(path):
    from tokenizers import Tokenizer, models, pre_tokenizers, processors, trainers

    tok = Tokenizer(models.BPE(unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    trainer = trainers.BpeTrainer(vocab_size=400, special_tokens=["[PAD]", "[CLS]", "[SEP]", "[UNK]", "[MASK]"])
    rng = np.random.default_rng(99)
    tok.train_from_iterator(make_corpus(rng, 200) + ["Where is the tower? What is tall! Who built it."], trainer)
    tok.post_processor = processors.TemplateProcessing(
        single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1",
        special_tokens=[("[CLS]", tok.token_to_id("[CLS]")), ("[SEP]", tok.token_to_id("[SEP]"))])
    tok.save(path)
    return tok


def hf_tokenizer(path):
    from transformers import PreTrainedTokenizerFast

    class Tok(PreTrainedTokenizerFast):
        def encode_plus(self, *a, **k):  # shim B
            return self.__call__(*a, **k)

    return Tok(tokenizer_file=path, cls_token="[CLS]", sep_token="[SEP]", pad_token="[PAD]", unk_token="[UNK]",
               mask_token="[MASK]")


TINY = dict(vocab_size=512, hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=192,
            pad_token_id=0, cls_token_id=1, sep_token_id=2)


def hf_tiny_model(weights, cls=None):
    import torch
    from transformers import ModernBertConfig, ModernBertModel

    hc = ModernBertConfig(max_position_embeddings=8192, bos_token_id=1, eos_token_id=2, **TINY)
    m = (cls or ModernBertModel)(hc).eval()
    return hc, m


def main():
    import torch

    _install_stubs()
    from oracle import modernbert_np as O

    out_json = {}
    rng = np.random.default_rng(1234)

    # ------------------------------------------------------------------ tokenizer
    tok_path = os.path.join(HERE, "tokenizer.json")
    build_tokenizer(tok_path)
    tok = hf_tokenizer(tok_path)
    assert tok.cls_token_id == 1 and tok.sep_token_id == 2 and tok.pad_token_id == 0

    # ------------------------------------------------------------------ a3 sentence split
    from verbatim_core.extractors import ModelSpanExtractor

    ext = ModelSpanExtractor.__new__(ModelSpanExtractor)
    texts = make_corpus(rng, 12) + [
        "", "   ", "No terminal punctuation", "One. Two!  Three?   Four", "Wait?! Really... Yes.",
        "Line one.\nLine two!\n\nLine three?", "Café ouvert. Ça va? Très bien!", "A.B.C. D", "Dr. Smith went. Home.",
        "Trailing space. ", " Leading. Space", "3.14 is pi. 2.71 is e.", "Emoji \U0001F600. Next!"]
    out_json["sentence_split"] = [{"text": t, "sentences": ext._split_into_sentences(t)} for t in texts]

    # ------------------------------------------------------------------ a4 packer
    from verbatim_core.extractor_models.dataset import Document, QADataset, QASample, Sentence

    cases = []
    questions = ["Where is the tower?", "Who built the iron bridge over the river in the old city?", "tall"]
    for qi, q in enumerate(questions):
        for t in texts[:8] + [texts[15], texts[17]]:
            sents = ext._split_into_sentences(t)
            if not sents:
                continue
            for max_len in (512, 40, 24):
                enc = QADataset.encode_question_and_sentences_with_offsets(
                    q, [Sentence(s, False, f"s{i}") for i, s in enumerate(sents)], tok, max_length=max_len)
                cases.append({"question": q, "sentences": sents, "max_length": max_len,
                              "input_ids": enc["input_ids"].tolist(),
                              "sentence_boundaries": [list(b) for b in enc["sentence_boundaries"]]})
    # a > 510-token chunk (overflow drop)
    long_sents = ext._split_into_sentences(" ".join(make_corpus(rng, 30)))
    enc = QADataset.encode_question_and_sentences_with_offsets(
        questions[1], [Sentence(s, False, f"s{i}") for i, s in enumerate(long_sents)], tok, max_length=512)
    cases.append({"question": questions[1], "sentences": long_sents, "max_length": 512,
                  "input_ids": enc["input_ids"].tolist(),
                  "sentence_boundaries": [list(b) for b in enc["sentence_boundaries"]]})
    out_json["packer"] = cases

    # ------------------------------------------------------------------ a8 encoder (transformers) + heads
    cfg = O.EncoderConfig(**TINY)
    w = O.random_weights(cfg, seed=7)
    from transformers import ModernBertForMaskedLM, ModernBertForTokenClassification

    hc, m = hf_tiny_model(w)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    npz = {}
    seq_rng = np.random.default_rng(21)
    for S in (7, 64, 130, 200):
        ids = seq_rng.integers(3, 400, size=S)
        with torch.no_grad():
            ref = m(input_ids=torch.tensor(ids)[None], output_hidden_states=True)
        npz[f"ids_{S}"] = ids.astype(np.int32)
        npz[f"hidden_{S}"] = ref.last_hidden_state[0].numpy()
        if S == 130:
            for l in range(5):  # hidden_states[l] = residual stream after l layers (l<4), [4] is post-final-norm
                if l < 4:
                    npz[f"resid_{S}_l{l}"] = ref.hidden_states[l][0].numpy()
    # right-padding invariance input: S=200 padded to 256 must equal unpadded (checked here)
    ids = npz["ids_200"]
    with torch.no_grad():
        padded = m(input_ids=torch.tensor(np.concatenate([ids, np.zeros(56, np.int64)]))[None],
                   attention_mask=torch.tensor([[1] * 200 + [0] * 56]))
    assert np.abs(padded.last_hidden_state[0, :200].numpy() - npz["hidden_200"]).max() < 1e-5

    # QA head through the reference's own QAModel.forward (shim A)
    from verbatim_core.extractor_models.model import QAModel

    head_rng = np.random.default_rng(5)
    Wc = (head_rng.standard_normal((2, 128)) * 128 ** -0.5).astype(np.float32)
    bc = (0.1 * head_rng.standard_normal(2)).astype(np.float32)
    classifier = torch.nn.Linear(128, 2)
    classifier.weight.data = torch.from_numpy(Wc)
    classifier.bias.data = torch.from_numpy(bc)
    ns = types.SimpleNamespace(bert=m, classifier=classifier)
    bounds = [(3, 20), (22, 60), (62, 129), (100, 400), (50, 10)]  # incl. end clamp (S=130) and an invalid range
    with torch.no_grad():
        preds = QAModel.forward(ns, torch.tensor(npz["ids_130"])[None], torch.ones(1, 130, dtype=torch.long), [bounds])
    npz["qa_Wc"], npz["qa_bc"] = Wc, bc
    npz["qa_bounds"] = np.asarray(bounds, dtype=np.int32)
    npz["qa_logits_130"] = preds[0].numpy()

    # token-classification head (HF) -- the arithmetic the v2 highlighter's model runs
    tc = ModernBertForTokenClassification(type(hc)(**{**hc.to_dict(), "num_labels": 2})).eval()
    sd = {"model." + k: torch.from_numpy(v) for k, v in w.items()}
    tk_rng = np.random.default_rng(6)
    tk = {"head.dense.weight": (tk_rng.standard_normal((128, 128)) * 0.05).astype(np.float32),
          "head.norm.weight": (1 + 0.1 * tk_rng.standard_normal(128)).astype(np.float32),
          "classifier.weight": (tk_rng.standard_normal((2, 128)) * 128 ** -0.5).astype(np.float32),
          "classifier.bias": (0.1 * tk_rng.standard_normal(2)).astype(np.float32)}
    sd.update({k: torch.from_numpy(v) for k, v in tk.items()})
    tc.load_state_dict(sd, strict=True)
    with torch.no_grad():
        npz["token_logits_200"] = tc(input_ids=torch.tensor(npz["ids_200"])[None]).logits[0].numpy()
    for k, v in tk.items():
        npz["tk_" + k] = v

    # MLM head (HF) -> SPLADE pooling restated (sentence-transformers absent: parity unpinned there)
    mlm = ModernBertForMaskedLM(hc).eval()
    sd = {"model." + k: torch.from_numpy(v) for k, v in w.items()}
    ml_rng = np.random.default_rng(8)
    ml = {"head.dense.weight": (ml_rng.standard_normal((128, 128)) * 0.08).astype(np.float32),
          "head.norm.weight": (1 + 0.1 * ml_rng.standard_normal(128)).astype(np.float32),
          "decoder.bias": (0.5 * ml_rng.standard_normal(512)).astype(np.float32)}
    sd.update({k: torch.from_numpy(v) for k, v in ml.items()})
    sd["decoder.weight"] = sd["model.embeddings.tok_embeddings.weight"]
    mlm.load_state_dict(sd, strict=True)
    with torch.no_grad():
        lg = mlm(input_ids=torch.tensor(npz["ids_64"])[None]).logits[0].numpy()
    npz["mlm_logits_64"] = lg
    npz["splade_row_64"] = np.log1p(np.maximum(lg, 0)).max(axis=0)
    for k, v in ml.items():
        npz["mlm_" + k] = v
    np.savez_compressed(os.path.join(HERE, "encoder_tiny.npz"), **npz)

    # ------------------------------------------------------------------ a1-a6 end to end through the reference extractor
    ext.model = types.SimpleNamespace(bert=m, classifier=classifier, config=hc)
    ext.model.__call__ = None

    class _M:  # callable wrapper: ModelSpanExtractor calls self.model(input_ids=..., ...)
        def __call__(self, input_ids, attention_mask, sentence_boundaries):
            return QAModel.forward(ns, input_ids, attention_mask, sentence_boundaries)

    ext.model = _M()
    ext.tokenizer = tok
    ext.device = "cpu"
    ext._torch = torch
    ext._format = ModelSpanExtractor._FORMAT_QA_MODEL
    ext.QADataset, ext.QASample, ext.DatasetDocument, ext.DatasetSentence = QADataset, QASample, Document, Sentence
    docs = make_corpus(np.random.default_rng(77), 10)
    results = [types.SimpleNamespace(text=t) for t in docs[:5] + ["", docs[0]]]
    e2e = []
    for thr in (0.5, 0.45, 0.55):
        ext.threshold = thr
        spans = ext.extract_spans("Where is the tower?", results)
        e2e.append({"threshold": thr, "question": "Where is the tower?", "texts": [r.text for r in results],
                    "spans": spans})
    # logits for margin checks
    lg_all = []
    for r in results:
        sents = ext._split_into_sentences(r.text)
        if not sents:
            lg_all.append([])
            continue
        enc = QADataset.encode_question_and_sentences_with_offsets(
            "Where is the tower?", [Sentence(s, False, "x") for s in sents], tok, max_length=512)
        with torch.no_grad():
            p = QAModel.forward(ns, enc["input_ids"][None], enc["attention_mask"][None], [enc["sentence_boundaries"]])
        lg_all.append(p[0].numpy().tolist())
    out_json["extract_e2e"] = {"runs": e2e, "logits": lg_all}

    # ------------------------------------------------------------------ a15 RRF
    from verbatim_rag.vector_stores.hybrid_search import (convert_hits_to_results, merge_hybrid_results,
                                                           normalize_weights, sanitize_hybrid_weights)

    def hits(ids):
        return [{"id": i, "distance": 1.0 - 0.01 * n, "entity": {"text": f"t{i}", "enhanced_text": f"e{i}",
                                                                "metadata": json.dumps({"document_id": f"d{i}"})}}
                for n, i in enumerate(ids)]

    rrf_cases = []
    for dense_ids, sparse_ids, ft_ids, weights, rrf_k, top_k in [
        (["a", "b", "c", "d"], ["c", "a", "e", "f"], None, {"dense": 0.5, "sparse": 0.5}, 60, 3),
        (["a", "b", "c"], ["d", "e", "f"], None, {"dense": 0.5, "sparse": 0.5}, 60, 6),      # all ties pairwise
        (["a", "b"], ["b", "a"], ["c", "a"], {"dense": 0.5, "sparse": 0.3, "full_text": 0.2}, 10, 3),
        (["x%d" % i for i in range(10)], ["x%d" % i for i in range(9, -1, -1)], None, {"dense": 2.0, "sparse": 1.0}, 60, 5),
        (["a", "b", "c"], [], None, {"dense": 0.0, "sparse": 0.0}, 60, 2),
    ]:
        rbm = {"dense": hits(dense_ids), "sparse": hits(sparse_ids)}
        if ft_ids is not None:
            rbm["full_text"] = hits(ft_ids)
        merged = merge_hybrid_results(rbm, top_k, weights, rrf_k=rrf_k)
        res = convert_hits_to_results(merged)
        rrf_cases.append({"dense": dense_ids, "sparse": sparse_ids, "full_text": ft_ids, "weights": weights,
                          "rrf_k": rrf_k, "top_k": top_k, "normalized": normalize_weights(rbm, weights),
                          "ids": [h["id"] for h in merged], "distances": [h["distance"] for h in merged],
                          "result_scores": [r.score for r in res], "result_metadata": [r.metadata for r in res]})
    out_json["rrf"] = rrf_cases
    out_json["sanitize"] = []
    for hw in ({"dense": 1, "sparse": 0.5, "bogus": 3}, {"dense": -1, "sparse": 2}, {"full_text": 1.0}):
        out_json["sanitize"].append({"in": hw, "out": sanitize_hybrid_weights(hw)})

    # ------------------------------------------------------------------ a16 highlights / citations
    from verbatim_core.response_builder import ResponseBuilder

    rb = ResponseBuilder()
    hl_cases = []
    for text, spans in [
        ("The cat sat on the mat.", ["cat"]),                                        # tests/test_response_builder.py:12-17
        ("The cat sat on the mat. The cat ran.", ["The cat", "cat sat", "mat"]),     # overlap suppression + repeats
        ("abc abc abc", ["abc", "bc a"]),
        ("Café \U0001F600 ouvert. Ça va? Très bien!", ["ouvert", "Très bien!"]),
        (docs[0], split[:2] if (split := ext._split_into_sentences(docs[0])) else []),
    ]:
        hs = rb._create_highlights(text, spans)
        hl_cases.append({"text": text, "spans": spans, "highlights": [h.model_dump() for h in hs]})
    out_json["highlights"] = hl_cases
    sr = [types.SimpleNamespace(text=t, metadata={"title": f"T{i}", "source": f"S{i}"}) for i, t in enumerate(docs[:3])]
    rel = {docs[0]: ext._split_into_sentences(docs[0])[:2], docs[1]: [], docs[2]: ext._split_into_sentences(docs[2])[:1]}
    resp = rb.build_response("Q?", "An answer.", sr, rel, display_span_count=2)
    out_json["build_response"] = {"texts": docs[:3], "relevant": rel, "display_span_count": 2,
                                  "response": resp.model_dump()}

    # ------------------------------------------------------------------ a12 VerbatimIndex.query dispatch trace
    from verbatim_rag.embedding_providers import DenseEmbeddingProvider, SparseEmbeddingProvider
    from verbatim_rag.index import VerbatimIndex
    from verbatim_rag.vector_stores.base import SearchResult, VectorStore

    calls = []

    class RecStore(VectorStore):
        enable_full_text = False

        def add_vectors(self, *a, **k):
            pass

        def query(self, **kw):
            calls.append({k: (v if not isinstance(v, (list, dict)) or k in ("hybrid_weights", "search_params") else "<vec>")
                          for k, v in kw.items()})
            return []

        def delete(self, ids):
            pass

    class D(DenseEmbeddingProvider):
        def embed_text(self, t): return [0.0, 1.0]
        def embed_batch(self, ts): return [[0.0, 1.0]] * len(ts)
        def get_dimension(self): return 2

    class Sp(SparseEmbeddingProvider):
        def embed_text(self, t): return {1: 0.5}
        def embed_batch(self, ts): return [{1: 0.5}] * len(ts)
        def get_dimension(self): return 10

    trace = []
    for name, d, s in (("both", D(), Sp()), ("dense", D(), None), ("sparse", None, Sp())):
        idx = VerbatimIndex(vector_store=RecStore(), dense_provider=d, sparse_provider=s)
        for kw in ({"text": "q", "k": 5}, {"text": "q", "k": 3, "search_type": "dense"}, {"text": None, "k": 4, "filter": "x"},
                   {"text": "q", "k": 5, "hybrid_weights": {"dense": 0.7, "sparse": 0.3}, "rrf_k": 10},
                   {"text": "q", "k": 2, "search_type": "sparse", "search_params": {"nprobe": 8}}):
            calls.clear()
            try:
                idx.query(**kw)
                trace.append({"providers": name, "kwargs": kw, "store_call": calls[-1] if calls else None})
            except Exception as e:
                trace.append({"providers": name, "kwargs": kw, "error": type(e).__name__})
    out_json["index_query_trace"] = trace

    # ------------------------------------------------------------------ a17 config-1 plumbing (static mode)
    from verbatim_rag.core import VerbatimRAG

    class CannedStore(RecStore):
        def query(self, **kw):
            calls.append(kw)
            k = kw.get("top_k", 5)
            return [SearchResult(id=f"c{i}", score=1.0 - 0.1 * i, metadata={"title": f"Doc {i}", "source": f"src{i}.md"},
                                 text=docs[i], enhanced_text=docs[i]) for i in range(k)]

    ext.threshold = 0.5
    rag = VerbatimRAG(index=VerbatimIndex(vector_store=CannedStore(), sparse_provider=Sp()), k=5, extractor=ext,
                      template_mode="static", llm_client=types.SimpleNamespace())
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):
        resp = rag.query("Where is the tower?")
    out_json["config1"] = {"question": "Where is the tower?", "docs": docs[:5], "threshold": 0.5,
                           "response": resp.model_dump()}

    with open(os.path.join(HERE, "host_fixtures.json"), "w") as f:
        json.dump(out_json, f, indent=1, ensure_ascii=True)

    # ------------------------------------------------------------------ base-size anchor (transformers fp32)
    from transformers import ModernBertConfig, ModernBertModel
    from verbatim_rag_amd.engine import ModernBertShape  # weight generator shared with tests/bench (data, not reference code)
    from verbatim_rag_amd.weights import random_init, random_qa_head

    shape = ModernBertShape.base()
    wb = random_init(shape, seed=1234)
    hb = ModernBertConfig()
    mb = ModernBertModel(hb).eval()
    mb.load_state_dict({k: torch.from_numpy(v) for k, v in wb.items()}, strict=True)
    qa_w, qa_b = random_qa_head(shape)
    brng = np.random.default_rng(2024)
    base = {}
    for n, S in enumerate((510, 333)):
        ids = np.concatenate([[shape.cls_token_id], brng.integers(1000, 50000, size=S - 1)])
        bnds, t = [], 26
        while t + 31 < S:
            bnds.append((t, t + 28))
            t += 30
        with torch.no_grad():
            hid = mb(input_ids=torch.tensor(ids)[None]).last_hidden_state[0]
            reprs = torch.stack([hid[a:b + 1].mean(0) for a, b in bnds])
            lg = reprs @ torch.from_numpy(qa_w).T + torch.from_numpy(qa_b)
        base[f"ids_{n}"] = ids.astype(np.int32)
        base[f"bounds_{n}"] = np.asarray(bnds, dtype=np.int32)
        base[f"logits_{n}"] = lg.numpy()
        base[f"hidden_rows_{n}"] = hid[:4].numpy()
        base[f"hidden_absmax_{n}"] = np.asarray([float(hid.abs().max())], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "encoder_base.npz"), **base)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
