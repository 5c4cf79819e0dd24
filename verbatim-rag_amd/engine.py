"""EncoderEngine: Python owner of one `vrag_encoder` handle (include/vrag_amd.h).

Takes HF-named fp32 weights (numpy arrays, e.g. read from safetensors) and a model
config, packs them onto the GPU through the C ABI and exposes the packed-batch entry
points.  All arithmetic happens inside libvrag_amd.so; this class only marshals
pointers.  It mirrors what `QAModel.from_pretrained(...).to(device).eval()` gives the
reference (packages/core/verbatim_core/extractors.py:176-181).
"""
from __future__ import annotations

import ctypes as C
import threading
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

_FP = C.POINTER(C.c_float)
_IP = C.POINTER(C.c_int32)


@dataclass
class ModernBertShape:
    """Architecture numbers of a ModernBERT checkpoint (config.json keys in comments)."""

    vocab_size: int = 50368            # vocab_size
    hidden_size: int = 768             # hidden_size
    num_hidden_layers: int = 22        # num_hidden_layers
    num_attention_heads: int = 12      # num_attention_heads
    intermediate_size: int = 1152      # intermediate_size
    global_attn_every_n_layers: int = 3
    local_attention: int = 128         # sliding window = local_attention // 2
    global_rope_theta: float = 160000.0
    local_rope_theta: float = 10000.0
    norm_eps: float = 1e-5
    pad_token_id: int = 50283
    cls_token_id: int = 50281
    sep_token_id: int = 50282

    @classmethod
    def base(cls) -> "ModernBertShape":
        return cls()

    @classmethod
    def large(cls) -> "ModernBertShape":
        return cls(hidden_size=1024, num_hidden_layers=28, num_attention_heads=16, intermediate_size=2624)

    @classmethod
    def from_hf_config(cls, cfg: dict) -> "ModernBertShape":
        g = cfg.get
        rp = g("rope_parameters") or {}
        return cls(
            vocab_size=g("vocab_size", 50368), hidden_size=g("hidden_size", 768),
            num_hidden_layers=g("num_hidden_layers", 22), num_attention_heads=g("num_attention_heads", 12),
            intermediate_size=g("intermediate_size", 1152),
            global_attn_every_n_layers=g("global_attn_every_n_layers", 3),
            local_attention=g("local_attention", 128),
            global_rope_theta=float(g("global_rope_theta", (rp.get("full_attention") or {}).get("rope_theta", 160000.0))),
            local_rope_theta=float(g("local_rope_theta", (rp.get("sliding_attention") or {}).get("rope_theta", 10000.0))),
            norm_eps=float(g("norm_eps", 1e-5)), pad_token_id=g("pad_token_id", 50283),
            cls_token_id=g("cls_token_id", 50281), sep_token_id=g("sep_token_id", 50282),
        )


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_FP)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def strip_prefix(weights: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Accepts QAModel names (`bert.*`, extractor_models/model.py:18,51), HF task-model names
    (`model.*`) or bare ModernBertModel names and returns bare names."""
    out = {}
    for k, v in weights.items():
        for p in ("bert.", "model."):
            if k.startswith(p):
                k = k[len(p):]
                break
        out[k] = v
    return out


class EncoderEngine:
    def __init__(
        self,
        shape: ModernBertShape,
        weights: Dict[str, np.ndarray],
        max_tokens: int = 8192,
        max_seqs: int = 64,
        max_seq_len: int = 512,
        max_ranges: int = 4096,
        micro_batch_tokens: int = 0,
        device: int = 0,
        operand_dtype: str = "bf16",
    ):
        """operand_dtype: type of the MFMA operands -- "bf16" (default: fp32's exponent range, sentence logits within
        3e-4 of the fp32 reference) or "f16" (11 significant bits at the same rate, saturating at 65504: per-token logits
        within 1e-3).  Accumulation, residual stream, LayerNorm, softmax, RoPE and heads are fp32 either way."""
        self._lib = _lib.load()
        _lib.require_gpu()
        if operand_dtype not in _lib.OPERAND_DTYPES:
            raise ValueError(f"operand_dtype must be one of {sorted(_lib.OPERAND_DTYPES)} (got {operand_dtype!r})")
        self.operand_dtype = operand_dtype
        self.shape = shape
        self._h = C.c_void_p()
        w = strip_prefix(weights)
        L = shape.num_hidden_layers
        keep: List[np.ndarray] = []

        def get(name):
            if name not in w:
                raise KeyError(f"weight '{name}' missing (have e.g. {list(w)[:4]})")
            a = _f32(w[name])
            keep.append(a)
            return a

        def arr(fmt, first=0):
            ptrs = (_FP * L)()
            for l in range(L):
                ptrs[l] = _fp(get(fmt.format(l))) if l >= first else None
            return ptrs

        cw = _lib.EncoderWeights()
        cw.tok_embeddings = _fp(get("embeddings.tok_embeddings.weight"))
        cw.emb_norm = _fp(get("embeddings.norm.weight"))
        cw.final_norm = _fp(get("final_norm.weight"))
        an, qkv, wo = arr("layers.{}.attn_norm.weight", 1), arr("layers.{}.attn.Wqkv.weight"), arr("layers.{}.attn.Wo.weight")
        mn, wi, wo2 = arr("layers.{}.mlp_norm.weight"), arr("layers.{}.mlp.Wi.weight"), arr("layers.{}.mlp.Wo.weight")
        cw.attn_norm, cw.wqkv, cw.wo, cw.mlp_norm, cw.wi, cw.wo_mlp = an, qkv, wo, mn, wi, wo2
        H, I, V = shape.hidden_size, shape.intermediate_size, shape.vocab_size
        expect = {
            "embeddings.tok_embeddings.weight": (V, H), "layers.0.attn.Wqkv.weight": (3 * H, H),
            "layers.0.attn.Wo.weight": (H, H), "layers.0.mlp.Wi.weight": (2 * I, H), "layers.0.mlp.Wo.weight": (H, I),
        }
        for k, shp in expect.items():
            if tuple(w[k].shape) != shp:
                raise ValueError(f"{k}: expected shape {shp}, got {tuple(w[k].shape)}")

        cfg = _lib.EncoderConfig(
            vocab_size=V, hidden_size=H, num_layers=L, num_heads=shape.num_attention_heads,
            intermediate_size=I, global_every=shape.global_attn_every_n_layers,
            sliding_window=shape.local_attention // 2, rope_theta_global=shape.global_rope_theta,
            rope_theta_local=shape.local_rope_theta, norm_eps=shape.norm_eps, pad_token_id=shape.pad_token_id,
            max_seq_len=max_seq_len, max_tokens=max_tokens, max_seqs=max_seqs, max_ranges=max_ranges,
            micro_batch_tokens=micro_batch_tokens, device=device, operand_dtype=_lib.OPERAND_DTYPES[operand_dtype],
        )
        self.max_tokens, self.max_seqs, self.max_seq_len, self.max_ranges = max_tokens, max_seqs, max_seq_len, max_ranges
        _lib.check("vrag_encoder_create", self._lib.vrag_encoder_create(C.byref(cfg), C.byref(cw), C.byref(self._h)))
        self._init_state()

    def _init_state(self) -> None:
        # load_batch -> run -> head -> read is a sequence on ONE workspace: every wrapper that drives this handle
        # (extractor, providers, reranker) holds this lock around its sequence, so two of them can share a handle.
        self.lock = threading.RLock()
        self.qa_labels = 0
        self.token_labels = 0
        self.has_mlm = False
        self._n_tokens = 0
        self._n_seqs = 0
        self._n_ranges = 0

    # ------------------------------------------------------------------ lifetime
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.vrag_encoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def graph_stats(self, enable: int = -1) -> Tuple[int, int]:
        """(graph replays so far, instantiated graphs) of the launch-bound path (`vrag_encoder_graph_stats`); `enable`:
        0 = eager launches only, > 0 = packed-row limit below which the layer schedule is captured and replayed."""
        n, c = C.c_int64(0), C.c_int32(0)
        _lib.check("vrag_encoder_graph_stats", self._lib.vrag_encoder_graph_stats(self._h, enable, C.byref(n), C.byref(c)))
        return int(n.value), int(c.value)

    def f16_saturated(self, reset: bool = True) -> bool:
        """True when an fp32 -> fp16 operand conversion on this device had to clamp to +-65504 since the last reset
        (`vrag_encoder_f16_saturated`): the logits computed meanwhile are not to be trusted -- use bf16 operands for
        that checkpoint.  Always False for bf16 handles' own work (they never convert to fp16).  Synchronises."""
        v = C.c_int32(0)
        _lib.check("vrag_encoder_f16_saturated", self._lib.vrag_encoder_f16_saturated(self._h, 1 if reset else 0, C.byref(v)))
        return bool(v.value)

    # ------------------------------------------------------------------ heads
    def set_qa_head(self, weight, bias) -> None:
        w, b = _f32(weight), _f32(bias)
        _lib.check("vrag_encoder_set_qa_head", self._lib.vrag_encoder_set_qa_head(self._h, _fp(w), _fp(b), w.shape[0]))
        self.qa_labels = int(w.shape[0])

    def set_token_head(self, dense_w, norm_w, cls_w, cls_b) -> None:
        d, n, w, b = _f32(dense_w), _f32(norm_w), _f32(cls_w), _f32(cls_b)
        _lib.check("vrag_encoder_set_token_head",
                   self._lib.vrag_encoder_set_token_head(self._h, _fp(d), _fp(n), _fp(w), _fp(b), w.shape[0]))
        self.token_labels = int(w.shape[0])

    def _set_head_precision(self, split_operands: bool) -> None:
        _lib.check("vrag_encoder_set_head_precision",
                   self._lib.vrag_encoder_set_head_precision(self._h, 1 if split_operands else 0))

    def set_mlm_head(self, dense_w, norm_w, decoder_b, decoder_w=None, split_operands: bool = True) -> None:
        """`split_operands` (default): both head GEMMs carry (value, remainder) operand pairs, SPLADE weights within 2e-3 of
        the fp32 reference; False = plain 16-bit operands, 3x less decoder work, weights within ~1e-2."""
        self._set_head_precision(split_operands)
        d, n = _f32(dense_w), _f32(norm_w)
        b = _f32(decoder_b) if decoder_b is not None else None
        dw = _f32(decoder_w) if decoder_w is not None else None
        _lib.check("vrag_encoder_set_mlm_head",
                   self._lib.vrag_encoder_set_mlm_head(self._h, _fp(d), _fp(n), _fp(dw) if dw is not None else None,
                                                       _fp(b) if b is not None else None))
        self.has_mlm = True

    # ------------------------------------------------------------------ batch
    def load_batch(self, sequences: Sequence[Sequence[int]], stream: Optional[int] = None) -> None:
        lens = _i32([len(s) for s in sequences])
        ids = _i32(np.concatenate([np.asarray(s, dtype=np.int32) for s in sequences])) if len(sequences) else _i32([])
        self.load_packed(ids, lens, stream)

    def load_packed(self, ids: np.ndarray, seq_lens: np.ndarray, stream: Optional[int] = None) -> None:
        ids, seq_lens = _i32(ids), _i32(seq_lens)
        _lib.check("vrag_encoder_load_batch",
                   self._lib.vrag_encoder_load_batch(self._h, ids.ctypes.data_as(_IP), seq_lens.ctypes.data_as(_IP),
                                                     len(seq_lens), stream))
        self._n_tokens, self._n_seqs, self._n_ranges = int(seq_lens.sum()), len(seq_lens), 0

    def load_ranges(self, seq_idx, start, end, stream: Optional[int] = None) -> None:
        s, a, b = _i32(seq_idx), _i32(start), _i32(end)
        _lib.check("vrag_encoder_load_ranges",
                   self._lib.vrag_encoder_load_ranges(self._h, s.ctypes.data_as(_IP), a.ctypes.data_as(_IP),
                                                      b.ctypes.data_as(_IP), len(s), stream))
        self._n_ranges = len(s)

    def run(self, stream: Optional[int] = None, n_layers: Optional[int] = None) -> None:
        if n_layers is None:
            _lib.check("vrag_encoder_run", self._lib.vrag_encoder_run(self._h, stream))
        else:
            _lib.check("vrag_encoder_run_layers", self._lib.vrag_encoder_run_layers(self._h, n_layers, stream))

    def run_qa_head(self, stream: Optional[int] = None) -> None:
        _lib.check("vrag_encoder_run_qa_head", self._lib.vrag_encoder_run_qa_head(self._h, stream))

    def read_qa_logits(self, stream: Optional[int] = None) -> np.ndarray:
        out = np.empty((self._n_ranges, self.qa_labels), dtype=np.float32)
        _lib.check("vrag_encoder_read_qa_logits", self._lib.vrag_encoder_read_qa_logits(self._h, _fp(out), stream))
        return out

    def run_pool(self, normalize: bool = True, stream: Optional[int] = None) -> None:
        _lib.check("vrag_encoder_run_pool", self._lib.vrag_encoder_run_pool(self._h, int(normalize), stream))

    def read_pool(self, stream: Optional[int] = None) -> np.ndarray:
        out = np.empty((self._n_ranges, self.shape.hidden_size), dtype=np.float32)
        _lib.check("vrag_encoder_read_pool", self._lib.vrag_encoder_read_pool(self._h, _fp(out), stream))
        return out

    def run_token_head(self, stream: Optional[int] = None) -> None:
        _lib.check("vrag_encoder_run_token_head", self._lib.vrag_encoder_run_token_head(self._h, stream))

    def read_token_logits(self, stream: Optional[int] = None) -> np.ndarray:
        out = np.empty((self._n_tokens, self.token_labels), dtype=np.float32)
        _lib.check("vrag_encoder_read_token_logits", self._lib.vrag_encoder_read_token_logits(self._h, _fp(out), stream))
        return out

    def run_splade(self, stream: Optional[int] = None) -> None:
        _lib.check("vrag_encoder_run_splade", self._lib.vrag_encoder_run_splade(self._h, stream))

    def read_splade(self, stream: Optional[int] = None) -> np.ndarray:
        out = np.empty((self._n_seqs, self.shape.vocab_size), dtype=np.float32)
        _lib.check("vrag_encoder_read_splade", self._lib.vrag_encoder_read_splade(self._h, _fp(out), stream))
        return out

    def read_splade_sparse(self, threshold: float = 0.0, cap_per_seq: int = 1024, stream: Optional[int] = None):
        """(counts[n_seqs], indices[n_seqs, cap], values[n_seqs, cap]) compacted on the device, vocabulary index
        ascending; raises VragError (status VRAG_ERR_CAPACITY = -3) when a row has more than cap_per_seq entries."""
        n = self._n_seqs
        counts = np.empty(n, dtype=np.int32)
        idx = np.empty((n, cap_per_seq), dtype=np.int32)
        val = np.empty((n, cap_per_seq), dtype=np.float32)
        _lib.check("vrag_encoder_read_splade_sparse", self._lib.vrag_encoder_read_splade_sparse(
            self._h, float(threshold), int(cap_per_seq), counts.ctypes.data_as(_IP), idx.ctypes.data_as(_IP), _fp(val), stream))
        return counts, idx, val

    def read_hidden(self, final_norm: bool = True, stream: Optional[int] = None) -> np.ndarray:
        out = np.empty((self._n_tokens, self.shape.hidden_size), dtype=np.float32)
        _lib.check("vrag_encoder_read_hidden", self._lib.vrag_encoder_read_hidden(self._h, int(final_norm), _fp(out), stream))
        return out

    # ------------------------------------------------------------------ one-call paths
    def qa_logits(self, sequences: Sequence[Sequence[int]],
                  boundaries: Sequence[Sequence[Tuple[int, int]]]) -> List[np.ndarray]:
        """[n_sent_i, labels] logits per sequence for inclusive (start, end) token ranges."""
        seq_idx, st, en = [], [], []
        for i, bs in enumerate(boundaries):
            for (s, e) in bs:
                seq_idx.append(i); st.append(s); en.append(e)
        if not seq_idx:
            return [np.zeros((0, self.qa_labels), np.float32) for _ in sequences]
        lens = _i32([len(s) for s in sequences])
        ids = _i32(np.concatenate([np.asarray(s, dtype=np.int32) for s in sequences]))
        si, sa, sb = _i32(seq_idx), _i32(st), _i32(en)
        out = np.empty((len(si), self.qa_labels), dtype=np.float32)
        _lib.check("vrag_encoder_extract_qa", self._lib.vrag_encoder_extract_qa(
            self._h, ids.ctypes.data_as(_IP), lens.ctypes.data_as(_IP), len(lens), si.ctypes.data_as(_IP),
            sa.ctypes.data_as(_IP), sb.ctypes.data_as(_IP), len(si), _fp(out)))
        self._n_tokens, self._n_seqs, self._n_ranges = int(lens.sum()), len(lens), len(si)
        res, o = [], 0
        for bs in boundaries:
            res.append(out[o:o + len(bs)])
            o += len(bs)
        return res

    def qa_logits_packed(self, ids: np.ndarray, seq_lens: np.ndarray, rng_seq: np.ndarray, rng_start: np.ndarray,
                         rng_end: np.ndarray) -> np.ndarray:
        """Flat form of `qa_logits`: concatenated ids + per-sequence lengths + flat inclusive ranges -> [n_ranges, labels]."""
        ids, lens = _i32(ids), _i32(seq_lens)
        si, sa, sb = _i32(rng_seq), _i32(rng_start), _i32(rng_end)
        out = np.empty((len(si), self.qa_labels), dtype=np.float32)
        if len(si) == 0:
            return out
        _lib.check("vrag_encoder_extract_qa", self._lib.vrag_encoder_extract_qa(
            self._h, ids.ctypes.data_as(_IP), lens.ctypes.data_as(_IP), len(lens), si.ctypes.data_as(_IP),
            sa.ctypes.data_as(_IP), sb.ctypes.data_as(_IP), len(si), _fp(out)))
        self._n_tokens, self._n_seqs, self._n_ranges = int(lens.sum()), len(lens), len(si)
        return out

    # ------------------------------------------------------------------ profiling
    def set_concurrency(self, n_streams: int) -> None:
        _lib.check("vrag_encoder_set_concurrency", self._lib.vrag_encoder_set_concurrency(self._h, int(n_streams)))

    def set_profiling(self, enabled) -> None:
        """False / 0 = off, True / 1 = HIP events around every launch, n > 1 = around every n-th launch of each kernel class."""
        _lib.check("vrag_encoder_set_profiling", self._lib.vrag_encoder_set_profiling(self._h, int(enabled)))

    def read_profile(self, reset: bool = True) -> Dict[str, Tuple[float, int]]:
        n = len(_lib.PROF_CLASSES)
        ms = (C.c_float * n)()
        cnt = (C.c_int64 * n)()
        _lib.check("vrag_encoder_read_profile", self._lib.vrag_encoder_read_profile(self._h, ms, cnt, int(reset)))
        return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(_lib.PROF_CLASSES)}


@dataclass
class BertShape:
    """Architecture numbers of a BERT / DistilBERT checkpoint (config.json keys in comments)."""

    vocab_size: int = 30522               # vocab_size
    hidden_size: int = 768                # hidden_size | dim
    num_hidden_layers: int = 12           # num_hidden_layers | n_layers
    num_attention_heads: int = 12         # num_attention_heads | n_heads
    intermediate_size: int = 3072         # intermediate_size | hidden_dim
    max_position_embeddings: int = 512
    norm_eps: float = 1e-12               # layer_norm_eps (DistilBERT hard-codes 1e-12)
    pad_token_id: int = 0
    cls_token_id: int = 101
    sep_token_id: int = 102
    model_type: str = "bert"              # "bert" | "distilbert"

    @classmethod
    def bert_base(cls) -> "BertShape":
        return cls()

    @classmethod
    def distilbert_base(cls) -> "BertShape":
        return cls(num_hidden_layers=6, model_type="distilbert")

    @classmethod
    def from_hf_config(cls, cfg: dict) -> "BertShape":
        g = cfg.get
        if g("model_type") == "distilbert":
            return cls(vocab_size=g("vocab_size", 30522), hidden_size=g("dim", 768), num_hidden_layers=g("n_layers", 6),
                       num_attention_heads=g("n_heads", 12), intermediate_size=g("hidden_dim", 3072),
                       max_position_embeddings=g("max_position_embeddings", 512), norm_eps=1e-12,
                       pad_token_id=g("pad_token_id", 0), model_type="distilbert")
        return cls(vocab_size=g("vocab_size", 30522), hidden_size=g("hidden_size", 768),
                   num_hidden_layers=g("num_hidden_layers", 12), num_attention_heads=g("num_attention_heads", 12),
                   intermediate_size=g("intermediate_size", 3072),
                   max_position_embeddings=g("max_position_embeddings", 512),
                   norm_eps=float(g("layer_norm_eps", 1e-12)), pad_token_id=g("pad_token_id", 0), model_type="bert")


class BertEncoderEngine(EncoderEngine):
    """Owner of a `vrag_encoder` handle built by `vrag_bert_encoder_create` (BERT / DistilBERT: post-LN, biased
    linears, learned positions, GELU MLP).  `weights` uses the flat names produced by
    `weights.bert_canonical(...)`; every batch entry point of `EncoderEngine` works unchanged."""

    def __init__(self, shape: BertShape, weights: Dict[str, np.ndarray], max_tokens: int = 8192, max_seqs: int = 64,
                 max_seq_len: int = 512, max_ranges: int = 4096, micro_batch_tokens: int = 0, device: int = 0,
                 operand_dtype: str = "bf16", mlm_split_operands: bool = True):
        self._lib = _lib.load()
        _lib.require_gpu()
        self._mlm_split_operands = bool(mlm_split_operands)
        if operand_dtype not in _lib.OPERAND_DTYPES:
            raise ValueError(f"operand_dtype must be one of {sorted(_lib.OPERAND_DTYPES)} (got {operand_dtype!r})")
        self.operand_dtype = operand_dtype
        self.shape = shape
        self._h = C.c_void_p()
        L, H, I = shape.num_hidden_layers, shape.hidden_size, shape.intermediate_size
        keep: List[np.ndarray] = []

        def get(name, shp=None):
            if name not in weights:
                raise KeyError(f"weight '{name}' missing (have e.g. {list(weights)[:4]})")
            a = _f32(weights[name])
            if shp is not None and tuple(a.shape) != tuple(shp):
                raise ValueError(f"{name}: expected shape {tuple(shp)}, got {tuple(a.shape)}")
            keep.append(a)
            return a

        def arr(fmt, shp):
            ptrs = (_FP * L)()
            for l in range(L):
                ptrs[l] = _fp(get(fmt.format(l), shp))
            return ptrs

        cw = _lib.BertWeights()
        cw.word_embeddings = _fp(get("emb.word", (shape.vocab_size, H)))
        cw.position_embeddings = _fp(get("emb.pos", (shape.max_position_embeddings, H)))
        cw.token_type_row = _fp(get("emb.type0", (H,))) if "emb.type0" in weights else None
        cw.emb_norm_w, cw.emb_norm_b = _fp(get("emb.ln.w", (H,))), _fp(get("emb.ln.b", (H,)))
        cw.wqkv, cw.bqkv = arr("l{}.wqkv", (3 * H, H)), arr("l{}.bqkv", (3 * H,))
        cw.wo, cw.bo = arr("l{}.wo", (H, H)), arr("l{}.bo", (H,))
        cw.attn_norm_w, cw.attn_norm_b = arr("l{}.ln1.w", (H,)), arr("l{}.ln1.b", (H,))
        cw.w1, cw.b1 = arr("l{}.w1", (I, H)), arr("l{}.b1", (I,))
        cw.w2, cw.b2 = arr("l{}.w2", (H, I)), arr("l{}.b2", (H,))
        cw.out_norm_w, cw.out_norm_b = arr("l{}.ln2.w", (H,)), arr("l{}.ln2.b", (H,))
        cfg = _lib.BertConfig(
            vocab_size=shape.vocab_size, hidden_size=H, num_layers=L, num_heads=shape.num_attention_heads,
            intermediate_size=I, max_position_embeddings=shape.max_position_embeddings, norm_eps=shape.norm_eps,
            pad_token_id=shape.pad_token_id, max_seq_len=min(max_seq_len, shape.max_position_embeddings),
            max_tokens=max_tokens, max_seqs=max_seqs, max_ranges=max_ranges, micro_batch_tokens=micro_batch_tokens,
            device=device, operand_dtype=_lib.OPERAND_DTYPES[operand_dtype])
        self.max_tokens, self.max_seqs, self.max_ranges = max_tokens, max_seqs, max_ranges
        self.max_seq_len = min(max_seq_len, shape.max_position_embeddings)
        _lib.check("vrag_bert_encoder_create", self._lib.vrag_bert_encoder_create(C.byref(cfg), C.byref(cw), C.byref(self._h)))
        self._init_state()
        self.pair_labels = 0
        if "emb.types" in weights:
            t = _f32(weights["emb.types"])
            _lib.check("vrag_encoder_set_token_types", self._lib.vrag_encoder_set_token_types(self._h, _fp(t), t.shape[0]))
        if "pooler.w" in weights and "cls.w" in weights:
            self.set_pair_head(weights["pooler.w"], weights["pooler.b"], weights["cls.w"], weights["cls.b"])
        if "mlm.dense.w" in weights:
            self.set_mlm_head_ex(weights["mlm.dense.w"], weights["mlm.dense.b"], weights["mlm.ln.w"], weights["mlm.ln.b"],
                                 weights.get("mlm.dec.b"), weights.get("mlm.dec.w"), split_operands=self._mlm_split_operands)

    def set_mlm_head_ex(self, dense_w, dense_b, norm_w, norm_b, decoder_b, decoder_w=None, split_operands: bool = True) -> None:
        self._set_head_precision(split_operands)
        opt = lambda a: _f32(a) if a is not None else None  # noqa: E731
        d, db, n, nb, b, dw = _f32(dense_w), opt(dense_b), _f32(norm_w), opt(norm_b), opt(decoder_b), opt(decoder_w)
        p = lambda a: _fp(a) if a is not None else None  # noqa: E731
        _lib.check("vrag_encoder_set_mlm_head_ex",
                   self._lib.vrag_encoder_set_mlm_head_ex(self._h, _fp(d), p(db), _fp(n), p(nb), p(dw), p(b)))
        self.has_mlm = True

    # ------------------------------------------------------------------ sentence pairs (cross-encoder reranking)
    def set_pair_head(self, pooler_w, pooler_b, cls_w, cls_b) -> None:
        pw, pb, cw, cb = _f32(pooler_w), _f32(pooler_b), _f32(cls_w), _f32(cls_b)
        _lib.check("vrag_encoder_set_pair_head",
                   self._lib.vrag_encoder_set_pair_head(self._h, _fp(pw), _fp(pb), _fp(cw), _fp(cb), cw.shape[0]))
        self.pair_labels = int(cw.shape[0])

    def load_token_types(self, type_ids: Sequence[Sequence[int]], stream: Optional[int] = None) -> None:
        """Per-token segment ids of the batch loaded last (same order and lengths as `load_batch`)."""
        flat = _i32(np.concatenate([np.asarray(t, dtype=np.int32) for t in type_ids]))
        if len(flat) != self._n_tokens:
            raise ValueError(f"{len(flat)} token types for a batch of {self._n_tokens} tokens")
        _lib.check("vrag_encoder_load_token_types",
                   self._lib.vrag_encoder_load_token_types(self._h, flat.ctypes.data_as(_IP), stream))

    def pair_logits(self, sequences: Sequence[Sequence[int]], type_ids: Sequence[Sequence[int]]) -> np.ndarray:
        """[n_seqs, labels] = classifier(tanh(pooler(h[CLS]))) for packed `[CLS] a [SEP] b [SEP]` sequences."""
        if not self.pair_labels:
            raise ValueError("engine has no pair head (pooler.* / classifier.* weights or set_pair_head)")
        self.load_batch(sequences)
        self.load_token_types(type_ids)
        self.run()
        _lib.check("vrag_encoder_run_pair_head", self._lib.vrag_encoder_run_pair_head(self._h, None))
        out = np.empty((self._n_seqs, self.pair_labels), dtype=np.float32)
        _lib.check("vrag_encoder_read_pair_logits", self._lib.vrag_encoder_read_pair_logits(self._h, _fp(out), None))
        return out
