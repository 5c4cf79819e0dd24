"""The SPLADE head at the vocabulary sizes the reference actually runs.

`SpladeProvider` hard-codes V = 30 522 (verbatim_rag/embedding_providers.py:117-133,168-169: BERT-family checkpoints such as
naver/splade-v3); a ModernBERT MLM head has V = 50 368.  Neither is a multiple of the 256-wide GEMM tile: V = 30 522 ends in a
58-column edge tile (119 full tiles), V = 50 368 in a 192-column one (196 full tiles) -- the column max of the fused epilogue
(csrc/gemm_bf16.hip, EPI_SPLADE) runs over zero-padded weight rows there.  Two layers are enough to exercise the head; the
encoders have the real widths (BERT-base 768 / 12 heads / 3072, ModernBERT-base 768 / 12 heads / 1152).

Two comparisons per case, sequences of 7, 130 and 512 tokens in one packed batch:
* head alone: oracle head (fp32 numpy: dense -> GELU -> LayerNorm -> decoder -> max_s log1p(relu)) applied to the hidden states
  the GPU encoder produced -- isolates the head GEMMs' operand rounding from the encoder's;
* end to end: the same rows against the oracle encoder + head.
Split operands (the default, include/vrag_amd.h vrag_encoder_set_head_precision) must hold 2e-3 (bf16) / 1e-3 (fp16) on the head;
plain operands are measured too and held to the looser bound the header states.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import bert_np as B  # noqa: E402
from oracle import modernbert_np as O  # noqa: E402

pytestmark = pytest.mark.gpu
LENS = (7, 130, 512)

# bounds on |row - oracle| : (head alone, end to end), ~4x above what `python tests/test_splade_real_vocab_gpu.py` measured on
# MI355X (profiles/r05_splade_real_vocab_probe.txt):
#   split operands  bf16: head 1.8e-5 / 1.3e-5, end to end 1.3e-2 (BERT, std-0.03 weights: the ENCODER's bf16 rounding) / 1.1e-3
#                   fp16: head 2.6e-6 / 2.7e-6, end to end 1.6e-3 / 1.5e-4
#   plain operands  bf16: head 1.06e-2 / 6.1e-3;  fp16: head 1.2e-3 / 7.6e-4
# i.e. with split operands the head is fp32-exact to 2e-5 (VERDICT r4 asked for 2e-3 / 1e-3) and what is left end to end is
# the encoder's own operand rounding, which the encoder tests bound.
BOUNDS = {
    ("bf16", True): (1e-4, 3e-2),
    ("f16", True): (2e-5, 4e-3),
    ("bf16", False): (3e-2, 4e-2),
    ("f16", False): (4e-3, 6e-3),
}
E2E_MODERNBERT = {"bf16": 4e-3, "f16": 6e-4}      # the ModernBERT case alone (trunc-normal 0.02 init): tighter end to end


def _bert_case(dtype, split):
    from verbatim_rag_amd.engine import BertEncoderEngine, BertShape

    cfg = B.BertConfig(vocab_size=30522, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=3072,
                       max_position_embeddings=512)
    W = B.random_weights(cfg, seed=11, kind="bert", std=0.03)
    shape = BertShape(vocab_size=cfg.vocab_size, hidden_size=768, num_hidden_layers=2, num_attention_heads=12,
                      intermediate_size=3072, max_position_embeddings=512, norm_eps=cfg.layer_norm_eps, pad_token_id=0,
                      cls_token_id=1, sep_token_id=2, model_type="bert")
    Wm = {k: v for k, v in W.items() if not k.startswith("mlm.")}
    eng = BertEncoderEngine(shape, Wm, max_tokens=2048, max_seqs=8, max_seq_len=512, max_ranges=16, operand_dtype=dtype)
    eng.set_mlm_head_ex(W["mlm.dense.w"], W["mlm.dense.b"], W["mlm.ln.w"], W["mlm.ln.b"], W["mlm.dec.b"], None,
                        split_operands=split)
    rng = np.random.default_rng(3)
    seqs = [rng.integers(3, cfg.vocab_size, size=n).astype(np.int32) for n in LENS]
    head = lambda hid: O.splade_pool(B.mlm_logits(cfg, W, hid))  # noqa: E731
    enc = lambda s: B.encoder_forward(cfg, W, s)  # noqa: E731
    return eng, seqs, head, enc, cfg.vocab_size, False


def _modernbert_case(dtype, split):
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape

    kw = dict(vocab_size=50368, hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=1152,
              pad_token_id=50283, cls_token_id=50281, sep_token_id=50282)
    cfg = O.EncoderConfig(**kw)
    w = O.random_weights(cfg, seed=21)
    rng = np.random.default_rng(5)
    Wd = O.trunc_normal(rng, (768, 768), 0.02)
    lnw = (1 + 0.1 * rng.standard_normal(768)).astype(np.float32)
    bdec = (0.3 * rng.standard_normal(cfg.vocab_size)).astype(np.float32)
    eng = EncoderEngine(ModernBertShape(**kw), w, max_tokens=2048, max_seqs=8, max_seq_len=512, max_ranges=16, operand_dtype=dtype)
    eng.set_mlm_head(Wd, lnw, bdec, None, split_operands=split)      # decoder tied to tok_embeddings (ModernBertForMaskedLM)
    seqs = [rng.integers(1000, 50000, size=n).astype(np.int32) for n in LENS]
    E = w["embeddings.tok_embeddings.weight"]
    head = lambda hid: O.splade_pool(O.mlm_logits(hid, Wd, lnw, E, bdec, cfg.norm_eps))  # noqa: E731
    enc = lambda s: O.encoder_forward(cfg, w, s)  # noqa: E731
    return eng, seqs, head, enc, cfg.vocab_size, True


def _measure(case, dtype, split):
    eng, seqs, head, enc, V, final_norm = case(dtype, split)
    try:
        eng.load_batch(seqs)
        eng.run()
        hid = eng.read_hidden(final_norm=final_norm)
        eng.run_splade()
        rows = eng.read_splade()
        counts, idx, val = eng.read_splade_sparse(0.0, cap_per_seq=V)
    finally:
        eng.close()
    assert rows.shape == (len(seqs), V) and (rows >= 0).all() and np.isfinite(rows).all()
    out = {"head": 0.0, "e2e": 0.0, "edge_head": 0.0, "support_head": 0, "support_e2e": 0}
    edge = V % 256
    o = 0
    for i, s in enumerate(seqs):
        ref_head = head(hid[o:o + len(s)])
        ref_e2e = head(enc(s))
        o += len(s)
        out["head"] = max(out["head"], float(np.abs(rows[i] - ref_head).max()))
        out["e2e"] = max(out["e2e"], float(np.abs(rows[i] - ref_e2e).max()))
        out["edge_head"] = max(out["edge_head"], float(np.abs(rows[i, V - edge:] - ref_head[V - edge:]).max()))
        # support (which terms are active) outside a band around zero of the head bound's width
        band = BOUNDS[(dtype, split)][0]
        out["support_head"] += int(((rows[i] > 0) != (ref_head > 0))[np.abs(ref_head) > band].sum())
        out["support_e2e"] += int(((rows[i] > 0) != (ref_e2e > 0))[np.abs(ref_e2e) > BOUNDS[(dtype, split)][1]].sum())
        # device compaction == dense row at this vocabulary
        nz = np.nonzero(rows[i] > 0)[0]
        assert counts[i] == len(nz) and np.array_equal(idx[i, :counts[i]], nz) and np.array_equal(val[i, :counts[i]], rows[i][nz])
        assert (ref_head[V - edge:] > 0).any()            # the edge tile carries active terms in this data
    return out


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("name,case", [("bert_v30522", _bert_case), ("modernbert_v50368", _modernbert_case)])
def test_splade_rows_at_real_vocabulary_split_operands(name, case, dtype):
    m = _measure(case, dtype, True)
    b_head, b_e2e = BOUNDS[(dtype, True)]
    assert m["head"] <= b_head and m["edge_head"] <= b_head, (name, dtype, m)
    assert m["e2e"] <= (E2E_MODERNBERT[dtype] if name.startswith("modernbert") else b_e2e), (name, dtype, m)
    assert m["support_head"] == 0 and m["support_e2e"] == 0, (name, dtype, m)


@pytest.mark.parametrize("name,case", [("bert_v30522", _bert_case), ("modernbert_v50368", _modernbert_case)])
def test_splade_rows_at_real_vocabulary_plain_operands(name, case):
    """The opt-out (a third of the decoder work): stated bound only."""
    m = _measure(case, "bf16", False)
    b_head, b_e2e = BOUNDS[("bf16", False)]
    assert m["head"] <= b_head and m["e2e"] <= b_e2e and m["support_head"] == 0, (name, m)


if __name__ == "__main__":   # probe: print what every mode achieves (the bounds above come from this)
    import json

    for name, case in (("bert_v30522", _bert_case), ("modernbert_v50368", _modernbert_case)):
        for dtype in ("bf16", "f16"):
            for split in (True, False):
                print(json.dumps({"case": name, "operands": dtype, "split": split, **_measure(case, dtype, split)}), flush=True)
