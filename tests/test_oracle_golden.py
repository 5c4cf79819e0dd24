"""Pins the oracle (oracle/modernbert_np.py) against golden vectors produced by `transformers`
and by the reference's own QAModel.forward (tests/golden/gen_golden.py)."""
import os

import numpy as np
import pytest

from oracle import modernbert_np as O

G = os.path.join(os.path.dirname(__file__), "golden")
TINY = dict(vocab_size=512, hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=192,
            pad_token_id=0, cls_token_id=1, sep_token_id=2)


@pytest.fixture(scope="module")
def tiny():
    cfg = O.EncoderConfig(**TINY)
    return cfg, O.random_weights(cfg, seed=7), np.load(os.path.join(G, "encoder_tiny.npz"))


@pytest.mark.parametrize("S", [7, 64, 130, 200])
def test_final_hidden_vs_transformers(tiny, S):
    cfg, w, z = tiny
    out = O.encoder_forward(cfg, w, z[f"ids_{S}"])
    assert np.abs(out - z[f"hidden_{S}"]).max() < 5e-6


def test_residual_stream_per_layer_vs_transformers(tiny):
    cfg, w, z = tiny
    _, hs = O.encoder_forward(cfg, w, z["ids_130"], return_all=True)
    for l in range(4):
        assert np.abs(hs[l] - z[f"resid_130_l{l}"]).max() < 5e-6, l


def test_sentence_head_vs_reference_qamodel(tiny):
    cfg, w, z = tiny
    hid = O.encoder_forward(cfg, w, z["ids_130"])
    bounds = [tuple(b) for b in z["qa_bounds"].tolist()]
    lg = O.qa_sentence_logits(hid, bounds, z["qa_Wc"], z["qa_bc"])
    assert lg.shape == z["qa_logits_130"].shape == (4, 2)  # the invalid (50,10) range is skipped, (100,400) clamped
    assert np.abs(lg - z["qa_logits_130"]).max() < 5e-6


def test_token_head_vs_transformers(tiny):
    cfg, w, z = tiny
    hid = O.encoder_forward(cfg, w, z["ids_200"])
    lg = O.token_logits(hid, z["tk_head.dense.weight"], z["tk_head.norm.weight"], z["tk_classifier.weight"],
                        z["tk_classifier.bias"], cfg.norm_eps)
    assert np.abs(lg - z["token_logits_200"]).max() < 1e-5


def test_mlm_and_splade_vs_transformers(tiny):
    cfg, w, z = tiny
    hid = O.encoder_forward(cfg, w, z["ids_64"])
    lg = O.mlm_logits(hid, z["mlm_head.dense.weight"], z["mlm_head.norm.weight"],
                      w["embeddings.tok_embeddings.weight"], z["mlm_decoder.bias"], cfg.norm_eps)
    assert np.abs(lg - z["mlm_logits_64"]).max() < 2e-5
    assert np.abs(O.splade_pool(lg) - z["splade_row_64"]).max() < 2e-5


def test_base_config_logits_vs_transformers():
    """ModernBERT-base shapes, weights from the shared seeded generator, fp32 transformers logits."""
    import verbatim_rag_amd  # noqa: F401
    from verbatim_rag_amd.engine import ModernBertShape
    from verbatim_rag_amd.weights import random_init, random_qa_head

    z = np.load(os.path.join(G, "encoder_base.npz"))
    shape = ModernBertShape.base()
    w = random_init(shape, seed=1234)
    qa_w, qa_b = random_qa_head(shape)
    cfg = O.EncoderConfig()
    ids, bounds = z["ids_1"], [tuple(b) for b in z["bounds_1"].tolist()]
    hid = O.encoder_forward(cfg, w, ids)
    assert np.abs(hid[:4] - z["hidden_rows_1"]).max() < 2e-4
    lg = O.qa_sentence_logits(hid, bounds, qa_w, qa_b)
    assert np.abs(lg - z["logits_1"]).max() < 5e-5


def test_blocked_topk_oracle_equals_the_scalar_statement():
    """oracle/topk_ref.c: the loop nest used at full size (16 queries per pass over a row, rows cut over the OpenMP
    threads, thread-private lists merged at the end) returns the scalar statement's bits -- arbitrary data, frequent
    exact ties (grid data), fewer hits than k, empty rows."""
    import numpy as np

    from oracle import topk_ref as T

    rng = np.random.default_rng(8)
    for n, dim, nq, k in [(5000, 96, 37, 10), (70, 8, 3, 64), (20000, 768, 17, 5)]:
        X = rng.standard_normal((n, dim)).astype(np.float32)
        Q = rng.standard_normal((nq, dim)).astype(np.float32)
        a, b = T.dense_topk(X, Q, k), T.dense_topk(X, Q, k, blocked=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        G = (rng.integers(-2, 3, (n, dim)) / 2).astype(np.float32)          # many exact ties: id order decides
        a, b = T.dense_topk(G, G[:nq], k), T.dense_topk(G, G[:nq], k, blocked=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    V, n = 500, 6000
    lens = rng.integers(0, 20, n)
    ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ix = np.concatenate([np.sort(rng.choice(V, int(m), replace=False)) for m in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    vv = rng.random(len(ix)).astype(np.float32)
    qlens = rng.integers(0, 9, 41)
    qp = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int64)
    qi = np.concatenate([np.sort(rng.choice(V, int(m), replace=False)) for m in qlens] + [np.zeros(0, np.int64)]).astype(np.int32)
    qv = rng.random(len(qi)).astype(np.float32)
    for k in (3, 64, 500):
        a, b = T.sparse_topk(ip, ix, vv, V, qp, qi, qv, k), T.sparse_topk(ip, ix, vv, V, qp, qi, qv, k, blocked=True)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
