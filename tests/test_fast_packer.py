"""The batched extractor's host path (cached `[SEP] sentence` tails + one searchsorted cut, vectorised threshold
selection) must hand the engine exactly what the general packer (packing.encode_question_and_sentences, the restatement
of dataset.py:127-243 pinned by the golden fixtures) would, and select the same sentences."""
import logging
import os
import types

import numpy as np
import pytest

import verbatim_rag_amd  # noqa: F401
from verbatim_rag_amd import packing
from verbatim_rag_amd.extractors import GpuModelSpanExtractor, select_sentences

G = os.path.join(os.path.dirname(__file__), "golden")

WORDS = ("alpha beta gamma delta epsilon zeta eta theta iota kappa lambda mu nu xi omicron pi rho sigma tau upsilon "
         "retrieval verbatim span extraction sentence boundary budget token question answer model index").split()


class RecordingEngine:
    """Engine stand-in: records what reaches the device entry point, returns logits that depend on the tokens."""
    max_seqs, max_tokens, max_ranges = 64, 8192, 256
    qa_labels = 2
    shape = types.SimpleNamespace()

    def __init__(self):
        self.sent = []

    def qa_logits_packed(self, ids, seq_lens, rng_seq, rng_start, rng_end):
        ids = np.asarray(ids)
        off = np.concatenate([[0], np.cumsum(seq_lens)])
        out = np.empty((len(rng_seq), 2), np.float32)
        for r, (s, a, b) in enumerate(zip(rng_seq, rng_start, rng_end)):
            seq = ids[off[s]:off[s + 1]]
            assert 0 <= a <= b < len(seq)
            v = float(seq[a:b + 1].astype(np.int64).sum() % 97) / 97.0
            out[r] = (0.5 - v, v - 0.5)
        for s in range(len(seq_lens)):
            m = np.asarray(rng_seq) == s
            self.sent.append((ids[off[s]:off[s + 1]].tolist(), list(zip(np.asarray(rng_start)[m].tolist(), np.asarray(rng_end)[m].tolist()))))
        return out


def _engine_logits(ids, bounds):
    out = np.empty((len(bounds), 2), np.float32)
    for r, (a, b) in enumerate(bounds):
        v = float(np.asarray(ids[a:b + 1], np.int64).sum() % 97) / 97.0
        out[r] = (0.5 - v, v - 0.5)
    return out


@pytest.fixture(scope="module")
def tokenizer():
    from tokenizers import Tokenizer

    return Tokenizer.from_file(os.path.join(G, "tokenizer.json"))


def _text(rng, n_sent, lo, hi):
    return " ".join(" ".join(rng.choice(WORDS, size=rng.integers(lo, hi))).capitalize() + "." for _ in range(n_sent))


@pytest.mark.parametrize("qa_max_length", [512, 64, 24])
def test_fast_path_equals_general_packer(tokenizer, qa_max_length, caplog):
    rng = np.random.default_rng(qa_max_length)
    eng = RecordingEngine()
    ext = GpuModelSpanExtractor(engine=eng, tokenizer=tokenizer, threshold=0.5, qa_max_length=qa_max_length)
    adapter = ext._tok
    questions = [_text(rng, 1, 3, 12)[:-1] + "?" for _ in range(6)] + ["", _text(rng, 1, 40, 60)]
    chunks_per_q = [[_text(rng, int(rng.integers(1, 14)), 2, 30) for _ in range(int(rng.integers(1, 6)))] + ["", "   "] for _ in questions]
    chunks_per_q[1].append(chunks_per_q[0][0])             # a cached chunk under another question
    with caplog.at_level(logging.ERROR):
        got = ext.extract_spans_batch(questions, [[types.SimpleNamespace(text=t) for t in cs] for cs in chunks_per_q])
    want_sent, budget = [], qa_max_length - 2
    for q, cs, res in zip(questions, chunks_per_q, got):
        q_ids = adapter.ids(q, add_special_tokens=True, max_length=budget)
        assert list(res.keys()) == list(dict.fromkeys(cs))
        for t in dict.fromkeys(cs):
            sents = packing.split_into_sentences(t)
            if not sents:
                assert res[t] == []
                continue
            smp = packing.encode_question_and_sentences(q_ids, adapter.ids_batch(sents, max_length=budget), adapter.sep_token_id,
                                                        max_length=qa_max_length)
            vb = packing.valid_boundaries(smp.sentence_boundaries, len(smp.input_ids))
            if not vb:
                assert res[t] == []
                continue
            assert res[t] == select_sentences(_engine_logits(smp.input_ids, vb), sents, 0.5), (q, t)
        for t in cs:                                          # device inputs in call order (duplicates are re-sent)
            sents = packing.split_into_sentences(t)
            if not sents:
                continue
            smp = packing.encode_question_and_sentences(q_ids, adapter.ids_batch(sents, max_length=budget), adapter.sep_token_id,
                                                        max_length=qa_max_length)
            vb = packing.valid_boundaries(smp.sentence_boundaries, len(smp.input_ids))
            if vb:
                want_sent.append((list(smp.input_ids), vb))
    assert eng.sent == want_sent


def test_budget_warning_is_kept(tokenizer, caplog):
    ext = GpuModelSpanExtractor(engine=RecordingEngine(), tokenizer=tokenizer, threshold=0.5, qa_max_length=32)
    rng = np.random.default_rng(3)
    with caplog.at_level(logging.WARNING):
        ext.extract_spans("what?", [types.SimpleNamespace(text=_text(rng, 6, 8, 12))])
    assert any("exceeded the 32-token budget; dropping" in r.getMessage() for r in caplog.records)


def test_sentence_without_tokens_takes_the_general_path(tokenizer):
    """A sentence that tokenises to nothing gives an (s, s-1) range, which QAModel skips so later rows shift
    (model.py:88-96): that case is left to the general packer + valid_boundaries."""
    ext = GpuModelSpanExtractor(engine=RecordingEngine(), tokenizer=tokenizer, threshold=0.5)
    entry = ext._cache_entry(["a.", "​", "b."], [[5], [], [7]])
    assert entry[4] is False and ext._pack_fast([1, 9, 2], entry) is None
    ok = ext._cache_entry(["a.", "b."], [[5], [7, 8]])
    ids, st, en = ext._pack_fast([ext._tok.cls_token_id, 9, ext._tok.sep_token_id], ok)
    sep = ext._tok.sep_token_id
    assert ids.tolist() == [ext._tok.cls_token_id, 9, sep, 5, sep, 7, 8, sep] and st.tolist() == [3, 5] and en.tolist() == [3, 6]


def test_two_handles_give_the_single_handle_result(tokenizer):
    rng = np.random.default_rng(11)
    questions = [_text(rng, 1, 3, 12)[:-1] + "?" for _ in range(40)]
    results = [[types.SimpleNamespace(text=_text(rng, int(rng.integers(2, 10)), 4, 20)) for _ in range(5)] for _ in questions]
    one = GpuModelSpanExtractor(engine=RecordingEngine(), tokenizer=tokenizer, threshold=0.5)
    a, b = RecordingEngine(), RecordingEngine()
    two = GpuModelSpanExtractor(engine=a, extra_engines=[b], tokenizer=tokenizer, threshold=0.5)
    assert two.extract_spans_batch(questions, results) == one.extract_spans_batch(questions, results)
    assert a.sent and b.sent and len(a.sent) + len(b.sent) == len(one.engine.sent)


class TokenHeadEngine:
    """Stand-in for the v2 (token-classification) engine: logits are a function of the token id alone."""
    max_seqs, max_tokens, max_ranges = 6, 700, 64
    token_labels, qa_labels = 2, 0
    shape = types.SimpleNamespace()

    def __init__(self):
        self.batches = 0

    def load_batch(self, seqs):
        self.ids = np.concatenate([np.asarray(s) for s in seqs])
        self.batches += 1

    def run(self):
        pass

    def run_token_head(self):
        pass

    def read_token_logits(self):
        v = (self.ids.astype(np.int64) * 7919 % 13) / 13.0
        return np.stack([0.5 - v, v - 0.5], 1).astype(np.float32)


def test_highlighter_cross_query_batching_host_logic(tokenizer):
    eng = TokenHeadEngine()
    ext = GpuModelSpanExtractor(engine=eng, tokenizer=tokenizer, model_format="highlighter", threshold=0.45, max_length=128,
                                doc_stride=16, min_span_chars=5, merge_gap_chars=3)
    ctxs = [" ".join([f"The tall iron tower number {i} in paris was built for the world fair."] * (5 + 7 * i)) for i in range(4)]
    qs = ["Where is the tower?", "Who built it?", "When was the fair?"]
    rs = [[types.SimpleNamespace(text=c) for c in ctxs], [types.SimpleNamespace(text=ctxs[2]), types.SimpleNamespace(text="")],
          [types.SimpleNamespace(text=c) for c in ctxs[::-1]]]
    got = ext.extract_spans_batch(qs, rs)
    n_batched = eng.batches
    want = [ext.extract_spans(q, r) for q, r in zip(qs, rs)]
    assert got == want and got[1][""] == [] and eng.batches - n_batched >= n_batched > 1
    assert all(s in c for d in got for c, spans in d.items() for s in spans) and any(v for d in got for v in d.values())


def test_native_packer_equals_the_python_fast_path(tokenizer):
    """`_pack_fast_many` (one `vrag_pack_qa_pairs` call per question: host code of the C ABI) against `_pack_fast` chunk by chunk:
    same ids, same inclusive ranges, None in the same places (question at the budget, empty sentence, nothing fits)."""
    rng = np.random.default_rng(8)
    for qa_max_length in (512, 96, 40, 12):
        ext = GpuModelSpanExtractor(engine=RecordingEngine(), tokenizer=tokenizer, threshold=0.5, qa_max_length=qa_max_length)
        texts = [_text(rng, int(rng.integers(1, 20)), 2, 40) for _ in range(40)] + ["", "x.", _text(rng, 1, 300, 400)]
        entries = ext._entries(texts) + [ext._cache_entry(["a.", "​", "b."], [[5], [], [7]])]
        for question in ("what?", "", _text(rng, 1, 6, 9), _text(rng, 1, 60, 90)):
            q_ids = ext._tok.ids(question, add_special_tokens=True, max_length=qa_max_length - 2)
            many = ext._pack_fast_many(q_ids, entries)
            assert len(many) == len(entries)
            for e, got in zip(entries, many):
                want = ext._pack_fast(q_ids, e) if e[0] else None
                assert (got is None) == (want is None), (qa_max_length, question[:20], e[0][:2])
                if want is not None:
                    assert got[0].dtype == np.int32 and got[1].dtype == np.int64
                    assert got[0].tolist() == want[0].tolist() and got[1].tolist() == want[1].tolist() and got[2].tolist() == want[2].tolist()


def test_native_packer_argument_and_capacity_errors():
    """`vrag_pack_qa_pairs` through ctypes: a full output buffer is an error status with a message (never a partial, silent
    result), null pointers are rejected, zero pairs is a no-op."""
    import ctypes as C

    from verbatim_rag_amd import _lib

    lib = _lib.load()
    ip, lp = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    q = np.asarray([1, 9, 9], np.int32)
    tail = np.asarray([2, 5, 6, 2, 7], np.int32)            # [SEP] 5 6 [SEP] 7
    cum = np.asarray([3, 5], np.int64)
    tails, cums = (C.c_uint64 * 1)(tail.ctypes.data), (C.c_uint64 * 1)(cum.ctypes.data)
    ng = np.asarray([2], np.int32)

    def call(ids_cap, rng_cap, n=1, q_ptr=None):
        ids, st, en = np.full(16, -1, np.int32), np.full(4, -1, np.int64), np.full(4, -1, np.int64)
        lens, kept, tot = np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(2, np.int64)
        rc = lib.vrag_pack_qa_pairs(q.ctypes.data_as(ip) if q_ptr is None else q_ptr, 3, n, tails, cums, ng.ctypes.data_as(ip), 510, 2,
                                    ids.ctypes.data_as(ip), ids_cap, st.ctypes.data_as(lp), en.ctypes.data_as(lp), rng_cap,
                                    lens.ctypes.data_as(ip), kept.ctypes.data_as(ip), tot.ctypes.data_as(lp))
        return rc, ids, st, en, lens, kept, tot

    rc, ids, st, en, lens, kept, tot = call(16, 4)
    assert rc == 0 and kept[0] == 2 and lens[0] == 9 and tot.tolist() == [9, 2]
    assert ids[:9].tolist() == [1, 9, 9, 2, 5, 6, 2, 7, 2] and st[:2].tolist() == [4, 7] and en[:2].tolist() == [5, 7]
    for ids_cap, rng_cap in ((8, 4), (16, 1)):
        rc = call(ids_cap, rng_cap)[0]
        assert rc == -3 and "capacity" in _lib.last_error()          # VRAG_ERR_CAPACITY
    assert call(16, 4, q_ptr=C.cast(None, ip))[0] == -1              # VRAG_ERR_INVALID
    rc, *_rest, tot = call(16, 4, n=0)
    assert rc == 0 and tot.tolist() == [0, 0]
