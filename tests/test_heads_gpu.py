"""Token-classification head, SPLADE head, dense pooling and the provider classes on the GPU."""
import os

import numpy as np
import pytest

from oracle import modernbert_np as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
TINY = dict(vocab_size=512, hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=192,
            pad_token_id=0, cls_token_id=1, sep_token_id=2)


@pytest.fixture(scope="module")
def setup():
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape

    cfg = O.EncoderConfig(**TINY)
    w = O.random_weights(cfg, seed=7)
    z = np.load(os.path.join(G, "encoder_tiny.npz"))
    eng = EncoderEngine(ModernBertShape(**TINY), w, max_tokens=8192, max_seqs=64, max_seq_len=2048, max_ranges=256)
    eng.set_token_head(z["tk_head.dense.weight"], z["tk_head.norm.weight"], z["tk_classifier.weight"], z["tk_classifier.bias"])
    eng.set_mlm_head(z["mlm_head.dense.weight"], z["mlm_head.norm.weight"], z["mlm_decoder.bias"])
    yield cfg, w, z, eng
    eng.close()


@pytest.mark.parametrize("dtype,tol", [("f16", 1e-3), ("bf16", 4e-3)])
def test_token_logits_vs_transformers_golden(setup, dtype, tol):
    """Per-token logits (for the v2 highlighter these ARE the span logits) against transformers'
    ModernBertForTokenClassification.  north_star's 1e-3 holds with fp16 MFMA operands (the highlighter extractor's
    default) and the split-operand head; bf16 operands (8 significant bits in every encoder GEMM) stay within 4e-3."""
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape

    cfg, w, z, _ = setup
    eng = EncoderEngine(ModernBertShape(**TINY), w, max_tokens=8192, max_seqs=64, max_seq_len=2048, max_ranges=256,
                        operand_dtype=dtype)
    eng.set_token_head(z["tk_head.dense.weight"], z["tk_head.norm.weight"], z["tk_classifier.weight"], z["tk_classifier.bias"])
    eng.load_batch([z["ids_200"], z["ids_64"]])
    eng.run()
    eng.run_token_head()
    lg = eng.read_token_logits()
    eng.close()
    assert lg.shape == (264, 2)
    ref = z["token_logits_200"]
    err = np.abs(lg[:200] - ref).max()
    assert err < tol, (dtype, err)
    p, pr = O.softmax_rows(lg[:200])[:, 1], O.softmax_rows(ref)[:, 1]
    assert np.abs(p - pr).max() < tol


def test_token_logits_base_depth_fp16_within_1e3():
    """The same bar at full depth (22 layers, H = 768: rounding accumulates with depth): random-init ModernBERT-base,
    fp16 operands, per-token logits vs the fp32 oracle within north_star's 1e-3; bf16 operands for comparison."""
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape
    from verbatim_rag_amd.weights import random_init

    shape = ModernBertShape.base()
    w = random_init(shape, seed=99)
    cfg = O.EncoderConfig()
    rng = np.random.default_rng(5)
    H = cfg.hidden_size
    Wd = O.trunc_normal(rng, (H, H), 0.02)
    lnw = (1 + 0.1 * rng.standard_normal(H)).astype(np.float32)
    Wc, bc = O.trunc_normal(rng, (2, H), 0.02), np.zeros(2, np.float32)
    seqs = [rng.integers(1000, 50000, size=n).astype(np.int32) for n in (256, 77)]
    ref = np.concatenate([O.token_logits(O.encoder_forward(cfg, w, s), Wd, lnw, Wc, bc, cfg.norm_eps) for s in seqs])
    errs = {}
    for dtype in ("f16", "bf16"):
        eng = EncoderEngine(shape, w, max_tokens=2048, max_seqs=8, max_seq_len=512, max_ranges=64, operand_dtype=dtype)
        eng.set_token_head(Wd, lnw, Wc, bc)
        eng.load_batch(seqs)
        eng.run()
        eng.run_token_head()
        errs[dtype] = float(np.abs(eng.read_token_logits() - ref).max())
        eng.close()
    assert errs["f16"] < 1e-3, errs
    assert errs["bf16"] < 1e-2, errs


def test_splade_rows_vs_golden_and_oracle(setup):
    cfg, w, z, eng = setup
    rng = np.random.default_rng(4)
    seqs = [z["ids_64"], rng.integers(3, 400, size=130).astype(np.int32), rng.integers(3, 400, size=9).astype(np.int32)]
    eng.load_batch(seqs)
    eng.run()
    hid = eng.read_hidden(final_norm=True)
    eng.run_splade()
    rows = eng.read_splade()
    assert rows.shape == (3, 512) and (rows >= 0).all()
    # the head alone (split operands, the default): oracle head on the hidden states the GPU encoder produced -- fp32-exact to
    # 1e-4 (measured 2e-5 at V = 30 522 / 50 368: tests/test_splade_real_vocab_gpu.py); the looser bounds below are the
    # ENCODER's bf16 operand rounding reaching the weights through a max over tokens
    o = 0
    for s, row in zip(seqs, rows):
        ref_h = O.splade_pool(O.mlm_logits(hid[o:o + len(s)], z["mlm_head.dense.weight"], z["mlm_head.norm.weight"],
                                           w["embeddings.tok_embeddings.weight"], z["mlm_decoder.bias"], cfg.norm_eps))
        o += len(s)
        assert np.abs(row - ref_h).max() < 1e-4, float(np.abs(row - ref_h).max())
        assert ((row > 0) == (ref_h > 0))[np.abs(ref_h) > 1e-4].all()
    assert np.abs(rows[0] - z["splade_row_64"]).max() < 2e-2
    for s, row in zip(seqs[1:], rows[1:]):
        hid = O.encoder_forward(cfg, w, s)
        lg = O.mlm_logits(hid, z["mlm_head.dense.weight"], z["mlm_head.norm.weight"],
                          w["embeddings.tok_embeddings.weight"], z["mlm_decoder.bias"], cfg.norm_eps)
        ref = O.splade_pool(lg)
        assert np.abs(row - ref).max() < 2e-2
        # the support (which terms are active) agrees except within the tolerance band around 0
        assert ((row > 0) == (ref > 0))[np.abs(ref) > 2e-2].all()


def test_splade_device_compaction_equals_dense_rows(setup):
    """vrag_encoder_read_splade_sparse: the same rows compacted on the GPU, index-ascending, both threshold rules;
    a capacity that is too small is reported, never truncated silently."""
    from verbatim_rag_amd._lib import VragError

    cfg, w, z, eng = setup
    rng = np.random.default_rng(8)
    seqs = [rng.integers(3, 400, size=n).astype(np.int32) for n in (64, 7, 130, 1)]
    eng.load_batch(seqs)
    eng.run()
    eng.run_splade()
    rows = eng.read_splade()
    for thr in (0.0, 1e-6, 0.05):
        counts, idx, val = eng.read_splade_sparse(thr, cap_per_seq=512)
        for i, row in enumerate(rows):
            nz = np.nonzero(row > thr)[0]
            assert counts[i] == len(nz), (thr, i, int(counts[i]), len(nz))
            assert np.array_equal(idx[i, :counts[i]], nz), (thr, i, idx[i, :8].tolist(), nz[:8].tolist())
            assert np.array_equal(val[i, :counts[i]], row[nz]), (thr, i)
    worst = int(max((r > 0).sum() for r in rows))
    with pytest.raises(VragError) as ei:
        eng.read_splade_sparse(0.0, cap_per_seq=max(1, worst - 1))
    assert ei.value.status == -3


def test_dense_pooling_vs_oracle(setup):
    cfg, w, z, eng = setup
    seqs = [z["ids_130"], z["ids_7"]]
    for mode in ("cls", "mean"):
        eng.load_batch(seqs)
        eng.load_ranges([0, 1], [0, 0], [0, 0] if mode == "cls" else [129, 6])
        eng.run()
        eng.run_pool(True)
        got = eng.read_pool()
        for s, g in zip(seqs, got):
            ref = O.dense_pool(O.encoder_forward(cfg, w, s), mode, True)
            assert abs(float(np.linalg.norm(g)) - 1.0) < 1e-5
            assert np.abs(g - ref).max() < 2e-3


def test_providers_contract(setup):
    from tokenizers import Tokenizer

    from verbatim_rag_amd.embedding_providers import GpuDenseProvider, GpuSpladeProvider

    cfg, w, z, eng = setup
    tok = Tokenizer.from_file(os.path.join(G, "tokenizer.json"))
    sp = GpuSpladeProvider(eng, tok)
    texts = ["Where is the tower?", "The iron bridge over the river.", "x"]
    one = sp.embed_text(texts[0])
    many = sp.embed_batch(texts)
    assert sp.get_dimension() == 512
    assert all(isinstance(k, int) and isinstance(v, float) for k, v in one.items())
    assert len(many) == 3 and {k for k, v in many[0].items() if abs(v) > 1e-6} == set(one)
    assert all(abs(many[0][k] - v) < 1e-6 for k, v in one.items())       # embed_text == embed_batch row (same kernels)
    dp = GpuDenseProvider(eng, tok, pooling="mean")
    v = dp.embed_text(texts[1])
    vb = dp.embed_batch(texts)
    assert dp.get_dimension() == 128 and len(v) == 128 and isinstance(v[0], float)
    assert np.allclose(v, vb[1], atol=1e-6) and abs(np.linalg.norm(v) - 1) < 1e-5


def test_highlighter_windows_long_context(setup):
    """v2 highlighter path: sliding windows + token head; spans are exact substrings."""
    from tokenizers import Tokenizer

    from verbatim_rag_amd.extractors import GpuModelSpanExtractor

    cfg, w, z, eng = setup
    tok = Tokenizer.from_file(os.path.join(G, "tokenizer.json"))
    ext = GpuModelSpanExtractor(engine=eng, tokenizer=tok, model_format="highlighter", threshold=0.5, max_length=256,
                                doc_stride=32, min_span_chars=5, merge_gap_chars=3)
    ctx = " ".join(["The tall iron tower in paris was built for the world fair."] * 40)
    import types

    out = ext.extract_spans("Where is the tower?", [types.SimpleNamespace(text=ctx), types.SimpleNamespace(text=" ")])
    assert set(out) == {ctx, " "} and out[" "] == []
    assert all(s in ctx and len(s) >= 5 for s in out[ctx])


def test_highlighter_cross_query_batching_equals_per_query_calls(setup):
    """extract_spans_batch on the v2 (token-classification) format: the windows of all queries share GPU batches and
    element i equals the single-query call."""
    import types

    from tokenizers import Tokenizer

    from verbatim_rag_amd.extractors import GpuModelSpanExtractor

    cfg, w, z, eng = setup
    tok = Tokenizer.from_file(os.path.join(G, "tokenizer.json"))
    ext = GpuModelSpanExtractor(engine=eng, tokenizer=tok, model_format="highlighter", threshold=0.45, max_length=128,
                                doc_stride=16, min_span_chars=5, merge_gap_chars=3)
    ctxs = [" ".join([f"The tall iron tower number {i} in paris was built for the world fair."] * (5 + 7 * i)) for i in range(4)]
    qs = ["Where is the tower?", "Who built it?", "When was the fair?"]
    rs = [[types.SimpleNamespace(text=c) for c in ctxs], [types.SimpleNamespace(text=ctxs[2]), types.SimpleNamespace(text="")],
          [types.SimpleNamespace(text=c) for c in ctxs[::-1]]]
    got = ext.extract_spans_batch(qs, rs)
    want = [ext.extract_spans(q, r) for q, r in zip(qs, rs)]
    assert got == want and list(got[1]) == [ctxs[2], ""] and got[1][""] == []
    assert any(len(v) > 0 for d in got for v in d.values())


def test_fp16_operands_report_saturation_instead_of_returning_clamped_logits(tmp_path, caplog):
    """fp16 conversions clamp at +-65504.  A checkpoint with outlier channels (here: three GeGLU channels of
    layer 1 scaled x3000, so gelu(x1) * x2 leaves fp16's range) must be REPORTED, not answered with plausible, wrong
    logits: `vrag_encoder_f16_saturated` is set, a healthy checkpoint leaves it clear, bf16 operands (fp32's exponent
    range) never touch it, and an extractor built from a model directory switches itself to bf16 and answers like an
    extractor that was constructed with bf16 operands."""
    import json
    import logging
    import shutil
    import types

    from safetensors.numpy import save_file

    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape
    from verbatim_rag_amd.extractors import GpuModelSpanExtractor

    cfg = O.EncoderConfig(**TINY)
    w = O.random_weights(cfg, seed=7)
    z = np.load(os.path.join(G, "encoder_tiny.npz"))
    hot = {k: v.copy() for k, v in w.items()}
    wi = hot["layers.1.mlp.Wi.weight"]                         # [2I, H]: rows 0..I-1 = x1 (gelu input), I..2I-1 = x2
    I = cfg.intermediate_size
    for c in (3, 77, 150):
        wi[c] *= 3000.0
        wi[I + c] *= 3000.0
    seqs = [z["ids_200"], z["ids_64"]]

    def run(weights, dtype):
        eng = EncoderEngine(ModernBertShape(**TINY), weights, max_tokens=8192, max_seqs=64, max_seq_len=2048, max_ranges=256,
                            operand_dtype=dtype)
        eng.set_token_head(z["tk_head.dense.weight"], z["tk_head.norm.weight"], z["tk_classifier.weight"], z["tk_classifier.bias"])
        eng.f16_saturated(reset=True)
        eng.load_batch(seqs)
        eng.run()
        eng.run_token_head()
        lg = eng.read_token_logits()
        flag = eng.f16_saturated(reset=True)
        again = eng.f16_saturated(reset=True)
        eng.close()
        return lg, flag, again

    lg, flag, again = run(w, "f16")
    assert not flag and np.isfinite(lg).all()
    lg_hot16, flag, again = run(hot, "f16")
    assert flag and not again                                  # reported once, cleared by the reset
    lg_hot, flag, _ = run(hot, "bf16")
    assert not flag and np.isfinite(lg_hot).all()
    assert np.abs(lg_hot16 - lg_hot).max() > 1e-2              # the clamped run really was wrong, not just flagged
    # the extractor built from a directory: v2 layout (model.* + head.* + classifier.*, auto_map naming a Highlighter)
    d = str(tmp_path)
    sd = {"model." + k: np.ascontiguousarray(v) for k, v in hot.items()}
    sd.update({"head.dense.weight": z["tk_head.dense.weight"], "head.norm.weight": z["tk_head.norm.weight"],
               "classifier.weight": z["tk_classifier.weight"], "classifier.bias": z["tk_classifier.bias"]})
    save_file(sd, os.path.join(d, "model.safetensors"))
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump({**TINY, "model_type": "modernbert", "max_position_embeddings": 8192, "global_attn_every_n_layers": 3,
                   "local_attention": 128, "global_rope_theta": 160000.0, "local_rope_theta": 10000.0, "norm_eps": 1e-5,
                   "auto_map": {"AutoModel": "modeling_highlighter.VerbatimHighlighterModel"}}, f)
    shutil.copy(os.path.join(G, "tokenizer.json"), os.path.join(d, "tokenizer.json"))
    kw = dict(threshold=0.5, max_length=256, doc_stride=32, min_span_chars=5, merge_gap_chars=3)
    ext16 = GpuModelSpanExtractor(model_path=d, **kw)
    ext_bf = GpuModelSpanExtractor(model_path=d, operand_dtype="bf16", **kw)
    assert ext16.engine.operand_dtype == "f16" and ext_bf.engine.operand_dtype == "bf16"
    ctx = " ".join(["The tall iron tower in paris was built for the world fair."] * 12)
    results = [types.SimpleNamespace(text=ctx), types.SimpleNamespace(text="A stone bridge crosses the river at night.")]
    with caplog.at_level(logging.WARNING):
        got = ext16.extract_spans("Where is the tower?", results)
    assert ext16.engine.operand_dtype == "bf16" and any("saturated" in r.message for r in caplog.records)
    assert got == ext_bf.extract_spans("Where is the tower?", results)
    for e in ext16.engines + ext_bf.engines:
        e.close()


def test_mlm_head_can_be_set_again_in_either_operand_form(setup):
    """`set_mlm_head` on an engine that already has a head (a BertEncoderEngine whose constructor attached the
    checkpoint's, or a caller swapping heads) must work -- the C setter releases the previous decoder images and rebuilds
    them in the form `vrag_encoder_set_head_precision` chose last (include/vrag_amd.h)."""
    cfg, w, z, eng = setup
    seqs = [z["ids_64"]]
    eng.load_batch(seqs)
    eng.run()
    eng.run_splade()
    first = eng.read_splade().copy()
    args = (z["mlm_head.dense.weight"], z["mlm_head.norm.weight"], z["mlm_decoder.bias"])
    eng.set_mlm_head(*args, split_operands=False)          # plain operands: ~1e-2 from the split form, not equal
    eng.run_splade()
    plain = eng.read_splade().copy()
    assert np.abs(plain - first).max() < 3e-2 and not np.array_equal(plain, first)
    eng.set_mlm_head(*args, split_operands=True)           # back: the same bits as the first head
    eng.run_splade()
    assert np.array_equal(eng.read_splade(), first)
