"""GPU-side sentence split (csrc/text.hip `vrag_split_sentences`) against the reference's arithmetic --
`re.split(r"(?<=[.!?])\s+", text)`, strip, drop empties (packages/core/verbatim_core/extractors.py:190-195; restated in
packing.split_into_sentences and pinned to the imported reference by tests/golden/host_fixtures.json:sentence_split)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
SPACES = ["\t", "\n", "\x0b", "\x0c", "\r", "\x1c", "\x1d", "\x1e", "\x1f", " ", "\x85", "\xa0", " ", " ", " ", " ",
          " ", " ", " ", " ", "　"]
NOT_SPACES = ["​", "⁠", "﻿", "᠎", "\xad", "⠀", "\x00", "\x1b", "\x7f", "\x86", "‎"]   # look-alikes str.isspace rejects


def test_captured_reference_cases_and_edges():
    from verbatim_rag_amd.packing import split_into_sentences, split_into_sentences_batch

    with open(os.path.join(G, "host_fixtures.json")) as f:
        cases = json.load(f)["sentence_split"]
    texts = [c["text"] for c in cases]
    assert split_into_sentences_batch(texts) == [c["sentences"] for c in cases]
    edges = ["", " ", ".", " . ", "a", "a.", "a. ", " a.", "a.b", "a. b", "a .b", "a . b", "a.  . b", "?! ?", "x!\n\ny?\t z",
             "end.　next line! \xa0nbsp? ​zero-width. ﻿bom", "é. ü! 中文。不分 割. 是? 的",
             "tab.\tafter", "no stop\nnew line", "trailing stop.", "   ", "\n.\n", "a." + " " * 300 + "b.",
             "🙂. 🙃! x", ". . . .", "a.\x1cb!\x1fc?\x85d"]
    assert split_into_sentences_batch(edges) == [split_into_sentences(t) for t in edges]
    assert split_into_sentences_batch([]) == []


def test_random_unicode_texts_equal_the_regex():
    from verbatim_rag_amd.packing import split_into_sentences, split_into_sentences_batch

    rng = np.random.default_rng(12)
    words = ["tower", "bridge", "é", "中文", "a", "Zürich", "x1", "🙂", "naïve", "…", "end"]
    pool = words + SPACES + NOT_SPACES + [".", "!", "?", ".", ". ", "? ", "!  ", ",", ";", "-"]
    texts = ["".join(rng.choice(pool, int(rng.integers(0, 120))).tolist()) for _ in range(3000)]
    texts.append(" ".join(f"Sentence number {i} ends here." for i in range(200)))      # more than `cap` sentences: host re-split
    got = split_into_sentences_batch(texts, cap=16)
    want = [split_into_sentences(t) for t in texts]
    assert got == want
    assert len(got[-1]) == 200


def test_extractor_ingest_uses_the_gpu_split_and_answers_like_the_host_split(monkeypatch):
    """`prepare_chunks` with an ingest-sized batch takes the GPU split; the chunk cache it fills equals the regex-filled one."""
    from tokenizers import Tokenizer

    from oracle import modernbert_np as O
    from verbatim_rag_amd import extractors as X
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape

    tiny = dict(vocab_size=512, hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=192,
                pad_token_id=0, cls_token_id=1, sep_token_id=2)
    eng = EncoderEngine(ModernBertShape(**tiny), O.random_weights(O.EncoderConfig(**tiny), seed=7), max_tokens=8192, max_seqs=64,
                        max_seq_len=512, max_ranges=1024)
    z = np.load(os.path.join(G, "encoder_tiny.npz"))
    eng.set_qa_head(z["qa_Wc"], z["qa_bc"])
    tok = Tokenizer.from_file(os.path.join(G, "tokenizer.json"))
    chunks = [f"The tall iron tower number {i} is in paris. It was built for the world fair!  Millions climb it? Yes." for i in range(300)]
    calls = []
    real = X.split_into_sentences_batch
    monkeypatch.setattr(X, "split_into_sentences_batch", lambda texts, device=0: (calls.append(len(texts)), real(texts, device))[1])
    a = X.GpuModelSpanExtractor(engine=eng, tokenizer=tok, threshold=0.5)
    a.prepare_chunks(chunks)
    assert calls == [300]
    b = X.GpuModelSpanExtractor(engine=eng, tokenizer=tok, threshold=0.5)
    monkeypatch.setattr(b, "GPU_SPLIT_MIN", 10 ** 9)
    b.prepare_chunks(chunks)
    assert calls == [300]
    for t in chunks:
        ea, eb = a._chunk_cache[t], b._chunk_cache[t]
        assert ea[0] == eb[0] and ea[1] == eb[1]
    eng.close()
