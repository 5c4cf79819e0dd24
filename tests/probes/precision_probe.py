#!/usr/bin/env python3
"""Precision study (CPU, torch fp32 arithmetic with emulated operand rounding): how many mantissa bits the MFMA operands
need for the v2 token-classification logits to stay within north_star's 1e-3 of the fp32 reference.  Mirrors
oracle/modernbert_np.py's forward with a rounding hook at every point where the HIP path rounds an MFMA operand
(LayerNorm output / h - c copy, q, k, v, P, attention output, GeGLU output, head input; weights).
  python tests/probes/precision_probe.py [n_seqs] [seq_len]
"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import modernbert_np as O  # noqa: E402

torch.set_num_threads(8)


def rnd(x, bits):
    """round-to-nearest-even to `bits` significant bits (24 = fp32: identity; 8 = bf16; 11 = fp16 without its range)."""
    if bits >= 24:
        return x
    if bits == 8:
        return x.to(torch.bfloat16).float()
    m, e = torch.frexp(x)
    return torch.ldexp(torch.round(m * (1 << bits)) / (1 << bits), e)


def ln(x, w, eps):
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    return xc / torch.sqrt((xc * xc).mean(-1, keepdim=True) + eps) * w


def forward(cfg, w, ids, act_bits, w_bits, hi_layers=0, hi_bits=24):
    """act_bits / w_bits: operand precision of layers < L - hi_layers; the last `hi_layers` layers use hi_bits for both."""
    S = len(ids)
    H, nh, d, I, L = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim, cfg.intermediate_size, cfg.num_hidden_layers
    t = lambda k: torch.from_numpy(w[k])   # noqa: E731
    h = ln(t("embeddings.tok_embeddings.weight")[torch.from_numpy(np.asarray(ids, np.int64))], t("embeddings.norm.weight"), cfg.norm_eps)
    pos = torch.arange(S)
    local = (pos[:, None] - pos[None, :]).abs() <= cfg.sliding_window
    tabs = {g: tuple(torch.from_numpy(x) for x in O.rope_tables(S, d, th)) for g, th in ((True, cfg.global_rope_theta), (False, cfg.local_rope_theta))}
    rot = lambda x: torch.cat([-x[..., d // 2:], x[..., : d // 2]], -1)   # noqa: E731
    for l in range(L):
        ab, wb = (hi_bits, hi_bits) if l >= L - hi_layers else (act_bits, w_bits)
        p = f"layers.{l}."
        a = h if l == 0 else ln(h, t(p + "attn_norm.weight"), cfg.norm_eps)
        qkv = (rnd(a, ab) @ rnd(t(p + "attn.Wqkv.weight"), wb).T).reshape(S, 3, nh, d)
        q, k, v = (qkv[:, i].transpose(0, 1) for i in range(3))
        cos, sin = tabs[cfg.is_global(l)]
        q = rnd((q * cos + rot(q) * sin) * d ** -0.5, ab)
        k = rnd(k * cos + rot(k) * sin, ab)
        v = rnd(v, ab)
        s = q @ k.transpose(1, 2)
        if not cfg.is_global(l):
            s = s.masked_fill(~local, float("-inf"))
        e = torch.exp(s - s.max(-1, keepdim=True).values)
        o = (rnd(e, ab) @ v) / e.sum(-1, keepdim=True)          # P rounded un-normalised, sum in fp32 (as the kernel does)
        o = rnd(o.transpose(0, 1).reshape(S, H), ab)
        h = h + o @ rnd(t(p + "attn.Wo.weight"), wb).T
        u = ln(h, t(p + "mlp_norm.weight"), cfg.norm_eps)
        x = rnd(u, ab) @ rnd(t(p + "mlp.Wi.weight"), wb).T
        act = rnd(torch.nn.functional.gelu(x[:, :I]) * x[:, I:], ab)
        h = h + act @ rnd(t(p + "mlp.Wo.weight"), wb).T
    return ln(h, t("final_norm.weight"), cfg.norm_eps)


def token_logits(hid, Wd, lnw, Wc, bc, eps, bits, wbits):
    x = torch.nn.functional.gelu(rnd(hid, bits) @ rnd(Wd, wbits).T)
    return ln(x, lnw, eps) @ Wc.T + bc


def main():
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    cfg = O.EncoderConfig()
    w = O.random_weights(cfg, seed=1234)
    rng = np.random.default_rng(0)
    H = cfg.hidden_size
    Wd = torch.from_numpy(O.trunc_normal(rng, (H, H), 0.02))
    lnw = torch.from_numpy((1 + 0.1 * rng.standard_normal(H)).astype(np.float32))
    Wc = torch.from_numpy(O.trunc_normal(rng, (2, H), 0.02))
    bc = torch.zeros(2)
    Wq = torch.from_numpy(O.trunc_normal(rng, (2, H), 0.02))
    seqs = [rng.integers(1000, 50000, size=S) for _ in range(n_seq)]
    bounds = [(25 + 30 * i, 25 + 30 * i + 28) for i in range(16)]
    cases = [("bf16 everywhere (round 1)", dict(act_bits=8, w_bits=8), 8, 8),
             ("bf16 body, head A-operand 16 bits", dict(act_bits=8, w_bits=8), 16, 8),
             ("bf16 body, head fp32", dict(act_bits=8, w_bits=8), 24, 24),
             ("last 2 layers + head fp32", dict(act_bits=8, w_bits=8, hi_layers=2), 24, 24),
             ("last 4 layers + head fp32", dict(act_bits=8, w_bits=8, hi_layers=4), 24, 24),
             ("last 8 layers + head fp32", dict(act_bits=8, w_bits=8, hi_layers=8), 24, 24),
             ("activations 11 bits (fp16-like), weights bf16", dict(act_bits=11, w_bits=8), 11, 8),
             ("activations 11 bits, weights 11 bits", dict(act_bits=11, w_bits=11), 11, 11),
             ("activations 16 bits (bf16 hi+lo), weights bf16", dict(act_bits=16, w_bits=8), 16, 8),
             ("activations 16 bits, weights 16 bits", dict(act_bits=16, w_bits=16), 16, 16),
             ("weights bf16 only", dict(act_bits=24, w_bits=8), 24, 8)]
    with torch.no_grad():
        refs = [forward(cfg, w, s, 24, 24) for s in seqs]
        ref_tok = [token_logits(r, Wd, lnw, Wc, bc, cfg.norm_eps, 24, 24) for r in refs]
        ref_sent = [torch.stack([r[a:b + 1].mean(0) for a, b in bounds]) @ Wq.T for r in refs]
        print(f"{'case':55s} {'hidden max':>11s} {'hidden mean':>11s} {'token-logit max':>15s} {'P(tok) max':>11s} {'sent-logit max':>14s}")
        for name, kw, hb, hwb in cases:
            hm = hmean = tm = pm = sm = 0.0
            for s, r, rt, rs in zip(seqs, refs, ref_tok, ref_sent):
                g = forward(cfg, w, s, **kw)
                d = (g - r).abs()
                hm, hmean = max(hm, d.max().item()), max(hmean, d.mean().item())
                gt = token_logits(g, Wd, lnw, Wc, bc, cfg.norm_eps, hb, hwb)
                tm = max(tm, (gt - rt).abs().max().item())
                pm = max(pm, (torch.softmax(gt, -1) - torch.softmax(rt, -1)).abs().max().item())
                gs = torch.stack([g[a:b + 1].mean(0) for a, b in bounds]) @ Wq.T
                sm = max(sm, (gs - rs).abs().max().item())
            print(f"{name:55s} {hm:11.2e} {hmean:11.2e} {tm:15.2e} {pm:11.2e} {sm:14.2e}", flush=True)


if __name__ == "__main__":
    main()
