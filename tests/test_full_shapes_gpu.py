"""GPU parity at the sizes BASELINE.json names (the other GPU tests use tiny geometries so the numpy oracle stays fast):
  * configs[1]: ModernBERT-base, 256 chunks x 512 tokens in ONE batch (two 65 536-token micro-batches on two streams,
    the bench's exact shape) -- sentence logits of a sample of chunks vs the fp32 oracle within north_star's 1e-3,
    and the batch result equals each sampled chunk run alone, bit for bit (padding-free packing leaks nothing);
  * configs[4]'s extractor: ModernBERT-large at FULL depth (28 layers, H = 1024, I = 2624 zero-padded to 2688),
    vocabulary cut to 4 096 rows to keep the host-side init small -- the encoder arithmetic is unaffected."""
import numpy as np
import pytest

from oracle import modernbert_np as O

pytestmark = pytest.mark.gpu


def _oracle_cfg(shape):
    return O.EncoderConfig(
        vocab_size=shape.vocab_size, hidden_size=shape.hidden_size, num_hidden_layers=shape.num_hidden_layers,
        num_attention_heads=shape.num_attention_heads, intermediate_size=shape.intermediate_size,
        global_attn_every_n_layers=shape.global_attn_every_n_layers, local_attention=shape.local_attention,
        global_rope_theta=shape.global_rope_theta, local_rope_theta=shape.local_rope_theta, norm_eps=shape.norm_eps,
        pad_token_id=shape.pad_token_id, cls_token_id=shape.cls_token_id, sep_token_id=shape.sep_token_id)


def test_configs1_batch_256x512_sample_vs_oracle():
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape
    from verbatim_rag_amd.weights import random_init, random_qa_head

    shape = ModernBertShape.base()
    w = random_init(shape, seed=1234)
    qa_w, qa_b = random_qa_head(shape)
    n, S, n_sent = 256, 512, 16
    rng = np.random.default_rng(1234)
    lens = [30] * 6 + [29] * 10
    seqs, bounds = [], []
    for _ in range(n):
        ids = [shape.cls_token_id] + rng.integers(1000, 50000, size=24).tolist()
        b = []
        for ln in lens:
            ids.append(shape.sep_token_id)
            b.append((len(ids), len(ids) + ln - 1))
            ids.extend(rng.integers(1000, 50000, size=ln).tolist())
        ids.append(shape.sep_token_id)
        assert len(ids) == S
        seqs.append(np.asarray(ids, np.int32))
        bounds.append(b)
    eng = EncoderEngine(shape, w, max_tokens=n * S, max_seqs=n, max_seq_len=S, max_ranges=n * n_sent, micro_batch_tokens=65536)
    eng.set_qa_head(qa_w, qa_b)
    got = eng.qa_logits(seqs, bounds)
    cfg = _oracle_cfg(shape)
    sample = [0, 127, 128, 255]          # first / last chunk of each micro-batch
    worst = 0.0
    for i in sample:
        ref = O.qa_sentence_logits(O.encoder_forward(cfg, w, seqs[i]), bounds[i], qa_w, qa_b)
        worst = max(worst, float(np.abs(got[i] - ref).max()))
        alone = eng.qa_logits([seqs[i]], [bounds[i]])[0]
        # a throughput-sized micro-batch runs the fused QKV + attention kernel, a lone chunk the two-kernel path (different
        # softmax bookkeeping, same operands): equal to rounding, and each within the oracle bar below
        assert np.abs(alone - got[i]).max() < 5e-4, f"chunk {i}: batch result differs from the chunk alone"
    eng.close()
    assert worst < 1e-3, worst


def test_ragged_throughput_batch_sample_vs_oracle():
    """BASELINE configs[1]'s model on the pair lengths real chunkers produce: 330 pairs of 1 .. 512 tokens (odd
    lengths included, mean ~200) in one 65 536-row micro-batch.  By the default rule this batch takes the fused QKV + attention
    kernel with WAVE-SLOT PACKING (several sequences per 512-token workgroup, groups packed best fit decreasing across the
    micro-batch: neighbours in a workgroup are not neighbours in the packed buffer); a sample of pairs -- the shortest, the
    longest, lengths around the 64-token slot edges -- is held to the oracle's 1e-3, and each equals itself extracted alone (the
    two-kernel path) to rounding."""
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape
    from verbatim_rag_amd.weights import random_init, random_qa_head

    shape = ModernBertShape.base()
    w = random_init(shape, seed=1234)
    qa_w, qa_b = random_qa_head(shape)
    rng = np.random.default_rng(77)
    lens = np.clip(64 + rng.gamma(2.0, 68.0, size=330), 40, 512).astype(int).tolist()
    lens[:8] = [1, 63, 64, 65, 128, 129, 511, 512]
    seqs, bounds = [], []
    for L in lens:
        ids = rng.integers(1000, 50000, size=L)
        ids[0] = shape.cls_token_id
        seqs.append(ids.astype(np.int32))
        step = max(1, L // 6)
        bounds.append([(a, min(L - 1, a + step - 1)) for a in range(0, L, step)][:8])
    eng = EncoderEngine(shape, w, max_tokens=sum(lens) + 8 * len(lens), max_seqs=len(lens), max_seq_len=512,
                        max_ranges=sum(len(b) for b in bounds), micro_batch_tokens=65536)
    eng.set_qa_head(qa_w, qa_b)
    assert sum(lens) > 40000                      # throughput-sized: above the launch-bound threshold
    got = eng.qa_logits(seqs, bounds)
    cfg = _oracle_cfg(shape)
    worst = 0.0
    for i in (0, 1, 2, 3, 5, 6, 7, 100, 329):
        ref = O.qa_sentence_logits(O.encoder_forward(cfg, w, seqs[i]), bounds[i], qa_w, qa_b)
        worst = max(worst, float(np.abs(got[i] - ref).max()))
        alone = eng.qa_logits([seqs[i]], [bounds[i]])[0]
        assert np.abs(alone - got[i]).max() < 5e-4, f"pair {i} ({lens[i]} tokens): batch result differs from the pair alone"
    eng.close()
    assert worst < 1e-3, worst


def test_modernbert_large_full_depth_vs_oracle():
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape
    from verbatim_rag_amd.weights import random_init, random_qa_head

    base = ModernBertShape.large()
    shape = ModernBertShape(**{**base.__dict__, "vocab_size": 4096, "pad_token_id": 0, "cls_token_id": 1, "sep_token_id": 2})
    assert shape.num_hidden_layers == 28 and shape.hidden_size == 1024
    w = random_init(shape, seed=77)
    qa_w, qa_b = random_qa_head(shape)
    rng = np.random.default_rng(78)
    seqs = [rng.integers(3, 4096, size=L).astype(np.int32) for L in (384, 129, 40)]
    bounds = [[(5, 60), (62, 200), (202, 383)], [(3, 50), (52, 128)], [(1, 39)]]
    eng = EncoderEngine(shape, w, max_tokens=2048, max_seqs=8, max_seq_len=512, max_ranges=64)
    eng.set_qa_head(qa_w, qa_b)
    got = eng.qa_logits(seqs, bounds)
    hid = eng.read_hidden(final_norm=True)
    eng.close()
    cfg = _oracle_cfg(shape)
    o = 0
    for s, b, g in zip(seqs, bounds, got):
        ref_h = O.encoder_forward(cfg, w, s)
        assert np.abs(hid[o:o + len(s)] - ref_h).max() < 5e-2
        assert np.abs(g - O.qa_sentence_logits(ref_h, b, qa_w, qa_b)).max() < 1e-3
        o += len(s)
