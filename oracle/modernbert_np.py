"""CPU restatement (numpy, fp32) of the ModernBERT forward + heads on the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py). Follows, line by line:

  * `transformers` 5.15.0 `models/modernbert/modeling_modernbert.py` (TF: below;
    the reference pins transformers==4.53.3 in pyproject.toml:28 and calls it at
    packages/core/verbatim_core/extractor_models/model.py:51,75):
      - embeddings  LN_nobias(E[ids])                         TF:52-71
      - MLP (GeGLU) Wo(gelu_erf(x1) * x2), x1,x2 = Wi(h).chunk(2)   TF:74-91
      - RoPE tables inv_freq = theta^(-2j/d), cat(freqs,freqs)      TF:141,150-163
      - rotate_half / apply_rotary_pos_emb (fp32)             TF:188-219
      - eager attention softmax_fp32(q k^T * d^-0.5 + mask) v TF:166-185
      - encoder layer: identity attn_norm on layer 0, pre-LN  TF:304-333
      - model forward, final_norm                             TF:434-478
      - prediction head LN(gelu(dense(h)))                    TF:481-490
      - MLM decoder (tied) / token classifier                 TF:499-550,660-699
    sliding mask |i-j| <= sliding_window (=local_attention//2)  TF:masking_utils.py:141-151
  * reference QAModel.forward sentence head (mean over inclusive token range,
    clamp end to S-1, skip invalid)  packages/core/verbatim_core/extractor_models/model.py:82-113
  * reference threshold select softmax(logits)[:,1] > thr
    packages/core/verbatim_core/extractors.py:270-277
  * SPLADE pooling max_s(log1p(relu(logits)) * mask): sentence-transformers 5.6.0
    SparseEncoder (absent here; restated, parity unpinned), call site
    verbatim_rag/embedding_providers.py:127-166.

All arithmetic is fp32 (numpy float32), matching the reference's CPU default.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import numpy as np

try:  # exact erf; scipy is in the image, fall back to math.erf vectorised
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf, otypes=[np.float64])

F32 = np.float32


@dataclass
class EncoderConfig:
    vocab_size: int = 50368
    hidden_size: int = 768
    num_hidden_layers: int = 22
    num_attention_heads: int = 12
    intermediate_size: int = 1152
    global_attn_every_n_layers: int = 3
    local_attention: int = 128          # sliding_window = local_attention // 2
    global_rope_theta: float = 160000.0
    local_rope_theta: float = 10000.0
    norm_eps: float = 1e-5
    pad_token_id: int = 50283
    cls_token_id: int = 50281
    sep_token_id: int = 50282

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def sliding_window(self) -> int:
        return self.local_attention // 2

    def is_global(self, layer: int) -> bool:
        # TF:configuration_modernbert.py:115-120  layer_types
        return layer % self.global_attn_every_n_layers == 0


def layer_norm(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    """nn.LayerNorm(bias=False): biased variance, eps inside sqrt (TF:61,70)."""
    x = x.astype(F32, copy=False)
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps))) * w.astype(F32)


def gelu_erf(x: np.ndarray) -> np.ndarray:
    """Exact GELU (ACT2FN['gelu'] == F.gelu default, erf form)."""
    x = x.astype(F32, copy=False)
    return (x * F32(0.5) * (F32(1.0) + _erf(x.astype(np.float64) / math.sqrt(2.0)).astype(F32))).astype(F32)


def rope_tables(seq_len: int, head_dim: int, theta: float) -> Tuple[np.ndarray, np.ndarray]:
    """cos/sin [S, head_dim] fp32, emb = cat(freqs, freqs) (TF:141,150-163)."""
    j = np.arange(0, head_dim, 2, dtype=F32)
    inv_freq = (F32(1.0) / (F32(theta) ** (j / F32(head_dim)))).astype(F32)
    pos = np.arange(seq_len, dtype=F32)
    freqs = pos[:, None] * inv_freq[None, :]
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb).astype(F32), np.sin(emb).astype(F32)


def rotate_half(x: np.ndarray) -> np.ndarray:
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


def encoder_forward(
    cfg: EncoderConfig,
    w: Dict[str, np.ndarray],
    input_ids: Sequence[int],
    return_all: bool = False,
):
    """One unpadded sequence, B=1 exactly as the reference calls it
    (packages/core/verbatim_core/extractors.py:260-268 -> model.py:75).

    `w` uses HF names without the model prefix:
      embeddings.tok_embeddings.weight, embeddings.norm.weight,
      layers.{i}.attn_norm.weight (absent for i=0), layers.{i}.attn.Wqkv.weight,
      layers.{i}.attn.Wo.weight, layers.{i}.mlp_norm.weight, layers.{i}.mlp.Wi.weight,
      layers.{i}.mlp.Wo.weight, final_norm.weight.
    Returns last_hidden_state [S, H] fp32 (and per-layer hidden states if asked).
    """
    ids = np.asarray(input_ids, dtype=np.int64)
    S = ids.shape[0]
    H, nh, d = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim
    h = layer_norm(w["embeddings.tok_embeddings.weight"][ids].astype(F32), w["embeddings.norm.weight"], cfg.norm_eps)
    hs = [h.copy()] if return_all else None

    pos = np.arange(S)
    dist = np.abs(pos[:, None] - pos[None, :])
    local_mask = dist <= cfg.sliding_window          # TF:masking_utils.py:146-149
    tabs = {
        True: rope_tables(S, d, cfg.global_rope_theta),
        False: rope_tables(S, d, cfg.local_rope_theta),
    }
    scale = F32(d ** -0.5)
    for l in range(cfg.num_hidden_layers):
        p = f"layers.{l}."
        a = h if l == 0 else layer_norm(h, w[p + "attn_norm.weight"], cfg.norm_eps)
        qkv = a @ w[p + "attn.Wqkv.weight"].astype(F32).T            # [S, 3H]
        qkv = qkv.reshape(S, 3, nh, d)
        q = qkv[:, 0].transpose(1, 0, 2)                               # [nh, S, d]
        k = qkv[:, 1].transpose(1, 0, 2)
        v = qkv[:, 2].transpose(1, 0, 2)
        cos, sin = tabs[cfg.is_global(l)]
        q = q * cos[None] + rotate_half(q) * sin[None]
        k = k * cos[None] + rotate_half(k) * sin[None]
        s = (q @ k.transpose(0, 2, 1)) * scale                         # [nh, S, S]
        if not cfg.is_global(l):
            s = np.where(local_mask[None], s, F32(-np.inf))
        s = s - s.max(axis=-1, keepdims=True)
        e = np.exp(s, dtype=F32)
        pr = e / e.sum(axis=-1, keepdims=True, dtype=F32)
        o = (pr @ v).transpose(1, 0, 2).reshape(S, H)
        h = h + o @ w[p + "attn.Wo.weight"].astype(F32).T
        u = layer_norm(h, w[p + "mlp_norm.weight"], cfg.norm_eps)
        x = u @ w[p + "mlp.Wi.weight"].astype(F32).T                   # [S, 2I]
        I = cfg.intermediate_size
        h = h + (gelu_erf(x[:, :I]) * x[:, I:]) @ w[p + "mlp.Wo.weight"].astype(F32).T
        h = h.astype(F32)
        if return_all:
            hs.append(h.copy())
    out = layer_norm(h, w["final_norm.weight"], cfg.norm_eps)
    if return_all:
        return out, hs
    return out


def qa_sentence_logits(
    hidden: np.ndarray, boundaries: Sequence[Tuple[int, int]], Wc: np.ndarray, bc: np.ndarray
) -> np.ndarray:
    """Reference QAModel.forward head (model.py:82-113): inclusive (start,end),
    end clamped to S-1, invalid ranges skipped (later rows shift up)."""
    S = hidden.shape[0]
    reprs = []
    for start, end in boundaries:
        if end >= S:
            end = S - 1
        if end < start or start < 0:
            continue
        reprs.append(hidden[start : end + 1].mean(axis=0, dtype=F32))
    if not reprs:
        return np.zeros((0, Wc.shape[0]), dtype=F32)
    r = np.stack(reprs).astype(F32)
    return (r @ Wc.astype(F32).T + bc.astype(F32)).astype(F32)


def softmax_rows(x: np.ndarray) -> np.ndarray:
    x = x.astype(F32)
    x = x - x.max(axis=-1, keepdims=True)
    e = np.exp(x, dtype=F32)
    return (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)


def select_sentences(logits: np.ndarray, sentences: List[str], threshold: float) -> List[str]:
    """extractors.py:270-277: strict `>`; index i reused against raw_sentences."""
    out = []
    if logits.shape[0] == 0:
        return out
    p = softmax_rows(logits)
    for i in range(p.shape[0]):
        if i < len(sentences) and p[i, 1] > threshold:
            out.append(sentences[i])
    return out


def prediction_head(hidden: np.ndarray, Wd: np.ndarray, ln_w: np.ndarray, eps: float) -> np.ndarray:
    """ModernBertPredictionHead: LN(gelu(dense(h))), dense has no bias (TF:481-490)."""
    return layer_norm(gelu_erf(hidden @ Wd.astype(F32).T), ln_w, eps)


def token_logits(hidden, Wd, ln_w, Wc, bc, eps: float) -> np.ndarray:
    """ModernBertForTokenClassification: classifier(head(h)) (TF:660-699)."""
    return (prediction_head(hidden, Wd, ln_w, eps) @ Wc.astype(F32).T + bc.astype(F32)).astype(F32)


def mlm_logits(hidden, Wd, ln_w, Wdec, bdec, eps: float) -> np.ndarray:
    """ModernBertForMaskedLM: decoder(head(h)); decoder.weight tied to tok_embeddings (TF:499-550)."""
    return (prediction_head(hidden, Wd, ln_w, eps) @ Wdec.astype(F32).T + bdec.astype(F32)).astype(F32)


def splade_pool(logits: np.ndarray) -> np.ndarray:
    """SPLADE max pooling over the (unpadded) tokens: max_s log1p(relu(logit))."""
    return np.log1p(np.maximum(logits.astype(F32), F32(0))).max(axis=0).astype(F32)


def dense_pool(hidden: np.ndarray, mode: str = "cls", normalize: bool = True) -> np.ndarray:
    """sentence-transformers Pooling (cls | mean) + Normalize (restated)."""
    v = hidden[0] if mode == "cls" else hidden.mean(axis=0, dtype=F32)
    v = v.astype(F32)
    if normalize:
        n = np.sqrt((v * v).sum(dtype=F32))
        v = v / np.maximum(n, F32(1e-12))
    return v.astype(F32)


# ----------------------------------------------------------------------------
# synthetic weights (same generator the product's random-init uses must NOT be
# shared: this one is the oracle's; tests feed the same arrays to both sides).
# ----------------------------------------------------------------------------
def trunc_normal(rng: np.random.Generator, shape, std: float, cutoff: float = 3.0) -> np.ndarray:
    x = rng.standard_normal(size=shape).astype(F32)
    bad = np.abs(x) > cutoff
    while bad.any():
        x[bad] = rng.standard_normal(size=int(bad.sum())).astype(F32)
        bad = np.abs(x) > cutoff
    return (x * F32(std)).astype(F32)


def random_weights(cfg: EncoderConfig, seed: int = 0, init_range: float = 0.02) -> Dict[str, np.ndarray]:
    """Random-init in the spirit of TF:352-400 (trunc-normal; 'in' std 0.02,
    'out' std 0.02/sqrt(2L)); LayerNorm weights jittered around 1 so that a
    kernel ignoring them is caught."""
    rng = np.random.default_rng(seed)
    H, I, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    out_std = init_range / math.sqrt(2.0 * L)
    w: Dict[str, np.ndarray] = {}
    w["embeddings.tok_embeddings.weight"] = trunc_normal(rng, (cfg.vocab_size, H), init_range)
    w["embeddings.norm.weight"] = (1.0 + 0.1 * rng.standard_normal(H)).astype(F32)
    for l in range(L):
        p = f"layers.{l}."
        if l > 0:
            w[p + "attn_norm.weight"] = (1.0 + 0.1 * rng.standard_normal(H)).astype(F32)
        w[p + "attn.Wqkv.weight"] = trunc_normal(rng, (3 * H, H), init_range)
        w[p + "attn.Wo.weight"] = trunc_normal(rng, (H, H), out_std)
        w[p + "mlp_norm.weight"] = (1.0 + 0.1 * rng.standard_normal(H)).astype(F32)
        w[p + "mlp.Wi.weight"] = trunc_normal(rng, (2 * I, H), init_range)
        w[p + "mlp.Wo.weight"] = trunc_normal(rng, (H, I), out_std)
    w["final_norm.weight"] = (1.0 + 0.1 * rng.standard_normal(H)).astype(F32)
    return w
