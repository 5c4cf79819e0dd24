"""CPU: the BERT-family oracle (oracle/bert_np.py) against golden vectors captured from `transformers`
BertForMaskedLM / DistilBertForMaskedLM (tests/golden/gen_golden_bert.py), and the product's weight-name
converter against the oracle's."""
import os

import numpy as np
import pytest

from oracle import bert_np as B
from oracle import modernbert_np as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    z = np.load(os.path.join(GOLD, f"{name}.npz"))
    V, H, L, NH, I, P = (int(x) for x in z["cfg"])
    cfg = B.BertConfig(vocab_size=V, hidden_size=H, num_hidden_layers=L, num_attention_heads=NH, intermediate_size=I,
                       max_position_embeddings=P)
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd:")}
    return z, cfg, sd


@pytest.mark.parametrize("name", ["bert_tiny", "distilbert_tiny"])
def test_oracle_matches_transformers(name):
    z, cfg, sd = _load(name)
    W = B.canonical_from_hf(sd)
    assert ("emb.type0" in W) == (name == "bert_tiny")
    for i in range(2):
        h = B.encoder_forward(cfg, W, z[f"ids{i}"])
        assert np.abs(h - z[f"hidden{i}"]).max() < 2e-5
        assert np.abs(B.mlm_logits(cfg, W, h) - z[f"mlm{i}"]).max() < 2e-5


@pytest.mark.parametrize("name", ["bert_tiny", "distilbert_tiny"])
def test_product_weight_converter_equals_oracle_converter(name):
    import verbatim_rag_amd  # noqa: F401
    from verbatim_rag_amd.weights import bert_canonical

    _, _, sd = _load(name)
    a, b = bert_canonical(sd), B.canonical_from_hf(sd)
    assert ("emb.types" in a) == (name == "bert_tiny")
    assert set(a) == set(b)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    # prefixed names (BertForMaskedLM) and bare names (BertModel) give the same encoder tensors
    bare = {k.split(".", 1)[1] if k.startswith(("bert.", "distilbert.")) else k: v for k, v in sd.items()}
    c = bert_canonical(bare)
    assert all(np.array_equal(a[k], c[k]) for k in a)


def test_layer_prefix_and_pooling_restatement():
    z, cfg, sd = _load("bert_tiny")
    W = B.canonical_from_hf(sd)
    h0 = B.encoder_forward(cfg, W, z["ids0"], n_layers=0)
    assert h0.shape == (len(z["ids0"]), cfg.hidden_size)
    # embedding LayerNorm output has (gain-weighted) zero mean / unit variance before gain and bias
    x = (h0 - W["emb.ln.b"]) / W["emb.ln.w"]
    assert np.abs(x.mean(-1)).max() < 1e-4 and np.abs(x.var(-1) - 1).max() < 1e-3
    rows = O.splade_pool(z["mlm0"])
    assert rows.shape == (cfg.vocab_size,) and (rows >= 0).all()
    v = O.dense_pool(z["hidden0"], "mean", True)
    assert abs(float(np.linalg.norm(v)) - 1.0) < 1e-6


def test_random_init_bert_shapes():
    import verbatim_rag_amd  # noqa: F401
    from verbatim_rag_amd.engine import BertShape
    from verbatim_rag_amd.weights import random_init_bert

    shp = BertShape(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                    max_position_embeddings=64)
    W = random_init_bert(shp, seed=1)
    cfg = B.BertConfig(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                       intermediate_size=256, max_position_embeddings=64)
    h = B.encoder_forward(cfg, W, [5, 6, 7, 8])
    assert h.shape == (4, 128) and np.isfinite(h).all()
    d = BertShape.from_hf_config({"model_type": "distilbert", "dim": 256, "n_layers": 3, "n_heads": 4, "hidden_dim": 512})
    assert (d.hidden_size, d.num_hidden_layers, d.model_type) == (256, 3, "distilbert")
    W2 = random_init_bert(d, seed=2)
    assert "emb.type0" not in W2


def test_pair_head_oracle_matches_transformers():
    z, cfg, sd = _load("bert_pair_tiny")
    W = B.canonical_from_hf(sd)
    import verbatim_rag_amd  # noqa: F401
    from verbatim_rag_amd.weights import bert_canonical

    P = bert_canonical(sd)
    assert set(P) == set(W) and all(np.array_equal(P[k], W[k]) for k in W)
    for i in range(3):
        h = B.encoder_forward(cfg, W, z[f"ids{i}"], type_ids=z[f"types{i}"])
        assert np.abs(h - z[f"hidden{i}"]).max() < 2e-5
        assert np.abs(B.pair_logits(cfg, W, h) - z[f"logits{i}"]).max() < 2e-6
