"""CPU: host logic of the GPU cross-encoder reranker -- pair packing against the `tokenizers` library's own
`longest_first` truncation, and the reference's ordering rule on a fake engine."""
import types

import numpy as np
import pytest
from tokenizers import Tokenizer
from tokenizers.models import WordLevel
from tokenizers.pre_tokenizers import Whitespace
from tokenizers.processors import TemplateProcessing

import verbatim_rag_amd  # noqa: F401
from verbatim_rag_amd.rerankers import BaseReranker, GpuCrossEncoderReranker, pack_pair
from verbatim_rag_amd.vector_stores import SearchResult


def _tok():
    vocab = {"[PAD]": 0, "[CLS]": 1, "[SEP]": 2, "[UNK]": 3}
    vocab.update({f"w{i}": 4 + i for i in range(200)})
    t = Tokenizer(WordLevel(vocab, unk_token="[UNK]"))
    t.pre_tokenizer = Whitespace()
    t.post_processor = TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1",
                                          special_tokens=[("[CLS]", 1), ("[SEP]", 2)])
    return t


@pytest.mark.parametrize("max_length", [8, 9, 16, 33, 64])
def test_pack_pair_equals_tokenizers_longest_first(max_length):
    rng = np.random.default_rng(max_length)
    t = _tok()
    t.enable_truncation(max_length=max_length, strategy="longest_first")
    plain = _tok()
    for _ in range(200):
        nq, nd = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        q = " ".join(f"w{int(i)}" for i in rng.integers(0, 200, nq))
        d = " ".join(f"w{int(i)}" for i in rng.integers(0, 200, nd))
        enc = t.encode(q, d)
        ids, tt = pack_pair(plain.encode(q, add_special_tokens=False).ids, plain.encode(d, add_special_tokens=False).ids,
                            1, 2, max_length)
        assert ids == enc.ids and tt == enc.type_ids, (nq, nd, max_length)


class FakeEngine:
    max_seq_len, max_seqs, max_tokens, pair_labels = 64, 4, 100, 1
    shape = types.SimpleNamespace(cls_token_id=1, sep_token_id=2)

    def __init__(self):
        self.batches = []

    def pair_logits(self, seqs, type_ids):
        self.batches.append(len(seqs))
        assert all(len(s) == len(t) for s, t in zip(seqs, type_ids))
        # score = number of document tokens equal to the first question token
        out = []
        for s, t in zip(seqs, type_ids):
            q0 = s[1]
            out.append([float(sum(1 for x, ty in zip(s, t) if ty == 1 and x == q0))])
        return np.asarray(out, dtype=np.float32)


def _res(i, text, score=0.0):
    return SearchResult(id=f"id{i}", score=score, metadata={}, text=text, enhanced_text=f"enh {text}")


def test_rerank_order_head_tail_and_batching():
    eng, tok = FakeEngine(), _tok()
    rr = GpuCrossEncoderReranker(eng, tok, rerank_k=6)
    texts = ["w1 w2", "w5 w5 w5", "w5", "w9", "w5 w5", "w5 w1 w5 w5 w5", "w5 w5 w5 w5 w5 w5 w5", "w0"]
    results = [_res(i, t, score=1.0 - 0.1 * i) for i, t in enumerate(texts)]
    out = rr.rerank("w5 what", results)
    assert [r.id for r in out[6:]] == ["id6", "id7"]                       # tail untouched (rerank_k)
    scores = rr.score("w5 what", texts[:6])
    assert [r.id for r in out[:6]] == [r.id for _, r in sorted(zip(scores, results[:6]), reverse=True)]
    assert out[0].id == "id5" and out[1].id == "id1"
    assert max(eng.batches) <= 4 and sum(eng.batches[:2]) == 6            # workspace-sized sub-batches
    assert rr.rerank("w5", []) == []
    assert isinstance(rr, BaseReranker)
    rr2 = GpuCrossEncoderReranker(eng, tok, text_field="enhanced_text")
    assert rr2._get_texts(results[:1]) == ["enh w1 w2"]
    with pytest.raises(ValueError):
        GpuCrossEncoderReranker(types.SimpleNamespace(pair_labels=0), tok)


def test_pipeline_applies_reranker_between_retrieval_and_extraction():
    from verbatim_rag_amd.pipeline import StaticVerbatimPipeline

    results = [_res(0, "Alpha one. Beta two."), _res(1, "Gamma three.")]

    class Index:
        def query(self, **kw):
            return list(results)

    class Extractor:
        def __init__(self):
            self.seen = None

        def extract_spans(self, question, rs):
            self.seen = [r.id for r in rs]
            return {r.text: [r.text.split(". ")[0].rstrip(".") + "."] if "." in r.text else [] for r in rs}

    class Reverse:
        def rerank(self, q, rs):
            return list(reversed(rs))

    class Broken:
        def rerank(self, q, rs):
            raise RuntimeError("down")

    ex = Extractor()
    resp = StaticVerbatimPipeline(Index(), ex, reranker=Reverse()).query("q")
    assert ex.seen == ["id1", "id0"] and resp.answer.index("Gamma") < resp.answer.index("Alpha")
    ex2 = Extractor()
    StaticVerbatimPipeline(Index(), ex2, reranker=Broken()).query("q")
    assert ex2.seen == ["id0", "id1"]
