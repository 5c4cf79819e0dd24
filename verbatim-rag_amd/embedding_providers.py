"""GPU embedding providers behind the reference's provider interfaces.

Kept: `DenseEmbeddingProvider` / `SparseEmbeddingProvider` ABCs and the exact return shapes
(verbatim_rag/embedding_providers.py:14-49): sparse `embed_text -> Dict[int,float]` keeping
|w| > 1e-6 (:138-146), `embed_batch -> List[Dict]` keeping exact non-zeros (:148-166), dense
`List[float]` rows (:73-77).  Replaced: sentence-transformers `SparseEncoder.encode` /
`SentenceTransformer.encode` (third-party, absent here) by the HIP encoder + fused SPLADE head
`max_s log1p(relu(mlm_logits))` / CLS-or-mean pooling + L2 normalise.  `engine` is an
`EncoderEngine` (ModernBERT backbone) or a `BertEncoderEngine` (BERT / DistilBERT: the checkpoints the
reference names -- `naver/splade-v3`, `opensearch-neural-sparse-encoding-doc-v2-distill`, bge-base;
embedding_providers.py:55,120; head_dim 64, or 32 as in the default dense model all-MiniLM-L6-v2).
"""
from __future__ import annotations

import threading
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from ._lib import VragError
from .packing import TokenizerAdapter


class DenseEmbeddingProvider(ABC):
    @abstractmethod
    def embed_text(self, text: str) -> List[float]:
        pass

    @abstractmethod
    def embed_batch(self, texts: List[str]) -> List[List[float]]:
        pass

    @abstractmethod
    def get_dimension(self) -> int:
        pass


class SparseEmbeddingProvider(ABC):
    @abstractmethod
    def embed_text(self, text: str) -> Dict[int, float]:
        pass

    @abstractmethod
    def embed_batch(self, texts: List[str]) -> List[Dict[int, float]]:
        pass

    @abstractmethod
    def get_dimension(self) -> int:
        pass


class _EncoderProvider:
    """`from_directory` builds the engine with **fp16 MFMA operands** (11 significant bits at the bf16 rate): embeddings feed an
    index whose top-k is held to bit-exactness, and nothing averages operand rounding away under SPLADE's max-pool -- measured end
    to end against the fp32 oracle, BERT-base width: SPLADE weights 1.6e-3 (fp16) vs 1.3e-2 (bf16), dense rows 6e-5 (fp16)
    (tests/test_splade_real_vocab_gpu.py, tests/test_e2e_text_in_gpu.py).  fp16 saturates at 65504 instead of overflowing: the
    library reports every clamp, and on the first report the provider rebuilds its engine with bf16 operands (fp32's exponent
    range) and runs the batch again -- `operand_dtype="bf16"` skips the probe; a provider that was HANDED its engine raises."""

    def __init__(self, engine: Any, tokenizer: Any, max_length: int = 512):
        self.engine = engine
        self.tokenizer = tokenizer
        self.max_length = min(max_length, engine.max_seq_len)
        self._tok = TokenizerAdapter.for_model(tokenizer, engine.shape)
        self._lock = getattr(engine, "lock", None) or threading.Lock()   # the handle's own lock: wrappers may share it
        self._rebuild_bf16 = None      # set by from_directory: () -> a bf16 engine of the same checkpoint

    def _f16_clamped(self) -> bool:
        """True = the engine was swapped for a bf16 one and the caller must run its batch again (caller holds the lock)."""
        eng = self.engine
        if getattr(eng, "operand_dtype", "bf16") != "f16" or not hasattr(eng, "f16_saturated") or not eng.f16_saturated(reset=True):
            return False
        if self._rebuild_bf16 is None:
            raise RuntimeError("fp16 MFMA operands saturated on this checkpoint (activations beyond 65504): "
                               "build the engine with operand_dtype='bf16'")
        import logging

        logging.getLogger(__name__).warning("fp16 MFMA operands saturated (activations beyond 65504): switching this provider to bf16 "
                                            "operands (construct it with operand_dtype='bf16' to skip the probe)")
        old, self.engine = self.engine, self._rebuild_bf16()
        self._rebuild_bf16 = None
        old.close()
        return True

    def _encode(self, texts: Sequence[str]) -> List[List[int]]:
        return [self._tok.ids(t, add_special_tokens=True, max_length=self.max_length) for t in texts]

    def _batches(self, seqs: List[List[int]]):
        start = 0
        while start < len(seqs):
            tok, end = 0, start
            while end < len(seqs) and end - start < self.engine.max_seqs and end - start < self.engine.max_ranges \
                    and tok + len(seqs[end]) <= self.engine.max_tokens:
                tok += len(seqs[end])
                end += 1
            if end == start:
                raise ValueError("a single text exceeds the engine workspace")
            yield start, end
            start = end


def load_encoder_directory(model_path: str, device: int = 0, max_tokens: int = 65536, max_seqs: int = 512,
                           max_seq_len: int = 512, splade_split_operands: bool = True, operand_dtype: str = "f16", **engine_kw):
    """(engine, tokenizer, raw config) from a local HF checkpoint directory -- the local-files counterpart of the model
    names the reference hands to sentence-transformers (`SpladeProvider(model_name)`, embedding_providers.py:117-133;
    `SentenceTransformersProvider(model_name)`, :52-71).  BERT / DistilBERT checkpoints get a `BertEncoderEngine` (MLM
    and pair heads attached when the tensors are there), ModernBERT checkpoints an `EncoderEngine` (+ MLM head).
    `splade_split_operands=False`: the MLM / SPLADE head with plain 16-bit operands (a third of the decoder work, weights within
    ~1e-2 of the fp32 head instead of 2e-5; include/vrag_amd.h vrag_encoder_set_head_precision)."""
    import json
    import os

    from . import engine as engine_mod
    from .weights import load_bert_safetensors_dir, load_safetensors_dir

    with open(os.path.join(model_path, "config.json")) as f:
        model_type = json.load(f).get("model_type")
    kw = dict(max_tokens=max_tokens, max_seqs=max_seqs, max_seq_len=max_seq_len, max_ranges=max(max_seqs, 64), device=device,
              operand_dtype=operand_dtype, **engine_kw)
    if model_type in ("bert", "distilbert"):
        shape, weights, cfg = load_bert_safetensors_dir(model_path)
        eng = engine_mod.BertEncoderEngine(shape, weights, mlm_split_operands=splade_split_operands, **kw)
    elif model_type == "modernbert":
        shape, tensors, cfg = load_safetensors_dir(model_path)
        eng = engine_mod.EncoderEngine(shape, tensors, **kw)
        if "head.dense.weight" in tensors and "decoder.bias" in tensors:        # ModernBertForMaskedLM: tied decoder
            eng.set_mlm_head(tensors["head.dense.weight"], tensors["head.norm.weight"], tensors["decoder.bias"],
                             tensors.get("decoder.weight"), split_operands=splade_split_operands)
    else:
        raise ValueError(f"{model_path}: model_type {model_type!r} is not bert / distilbert / modernbert")
    try:
        from transformers import AutoTokenizer

        tokenizer = AutoTokenizer.from_pretrained(model_path)
    except Exception:
        from tokenizers import Tokenizer

        tokenizer = Tokenizer.from_file(os.path.join(model_path, "tokenizer.json"))
    return eng, tokenizer, cfg


def _st_pooling_mode(model_path: str, default: str = "cls") -> str:
    """sentence-transformers checkpoints say how they pool in `1_Pooling/config.json` (bge: CLS, MiniLM: mean)."""
    import json
    import os

    try:
        with open(os.path.join(model_path, "1_Pooling", "config.json")) as f:
            pc = json.load(f)
    except OSError:
        return default
    if pc.get("pooling_mode_mean_tokens"):
        return "mean"
    if pc.get("pooling_mode_cls_token"):
        return "cls"
    raise ValueError(f"{model_path}: only CLS and mean pooling are implemented (1_Pooling/config.json: {pc})")


class GpuSpladeProvider(_EncoderProvider, SparseEmbeddingProvider):
    """SpladeProvider (embedding_providers.py:117-169) on the HIP encoder + fused SPLADE head."""

    def __init__(self, engine: Any, tokenizer: Any, max_length: int = 512, sparse_cap: int = 1024):
        super().__init__(engine, tokenizer, max_length)
        self.sparse_cap = int(sparse_cap)
        if not engine.has_mlm:
            raise ValueError("engine has no MLM head (EncoderEngine.set_mlm_head)")

    @classmethod
    def from_directory(cls, model_path: str, device: int = 0, max_length: int = 512, operand_dtype: str = "f16", **kw) -> "GpuSpladeProvider":
        """`SpladeProvider(model_name, device)` (embedding_providers.py:120-133) for a checkpoint on disk."""
        engine, tokenizer, _cfg = load_encoder_directory(model_path, device=device, max_seq_len=max_length, operand_dtype=operand_dtype)
        self = cls(engine, tokenizer, max_length=max_length, **kw)
        if operand_dtype == "f16":
            self._rebuild_bf16 = lambda: load_encoder_directory(model_path, device=device, max_seq_len=max_length, operand_dtype="bf16")[0]
        return self

    def _rows(self, texts: Sequence[str]) -> np.ndarray:
        seqs = self._encode(texts)
        out = np.empty((len(seqs), self.engine.shape.vocab_size), np.float32)
        with self._lock:
            for a, b in self._batches(seqs):
                while True:
                    self.engine.load_batch(seqs[a:b])
                    self.engine.run()
                    self.engine.run_splade()
                    out[a:b] = self.engine.read_splade()
                    if not self._f16_clamped():
                        break
        return out

    def _dicts(self, texts: Sequence[str], threshold: float) -> List[Dict[int, float]]:
        """Rows compacted on the GPU (vocabulary order = np.nonzero order); a row with more than `sparse_cap`
        entries (e.g. an untrained model) falls back to the dense read for that batch."""
        seqs = self._encode(texts)
        out: List[Dict[int, float]] = []
        with self._lock:
            for a, b in self._batches(seqs):
                while True:
                    self.engine.load_batch(seqs[a:b])
                    self.engine.run()
                    self.engine.run_splade()
                    part: List[Dict[int, float]] = []
                    try:
                        counts, idx, val = self.engine.read_splade_sparse(threshold, self.sparse_cap)
                        for i in range(b - a):
                            n = int(counts[i])
                            part.append(dict(zip(idx[i, :n].tolist(), val[i, :n].tolist())))
                    except VragError as exc:
                        if exc.status != -3:   # VRAG_ERR_CAPACITY
                            raise
                        rows = self.engine.read_splade()
                        for row in rows:
                            nz = np.nonzero(row > threshold)[0]
                            part.append({int(i): float(row[i]) for i in nz})
                    if not self._f16_clamped():
                        break
                out.extend(part)
        return out

    def embed_text(self, text: str) -> Dict[int, float]:
        return self._dicts([text], 1e-6)[0]                # |w| > 1e-6, embedding_providers.py:141-145 (w >= 0)

    def embed_batch(self, texts: List[str]) -> List[Dict[int, float]]:
        return self._dicts(texts, 0.0)                     # every non-zero, embedding_providers.py:161-163

    def embed_queries(self, texts: List[str]) -> List[Dict[int, float]]:
        """`[embed_text(t) for t in texts]` (the |w| > 1e-6 rule) as shared device batches -- the query side of a
        cross-query batch; `embed_batch` is the ingest side and keeps every non-zero."""
        return self._dicts(texts, 1e-6)

    def get_dimension(self) -> int:
        return int(self.engine.shape.vocab_size)


class GpuDenseProvider(_EncoderProvider, DenseEmbeddingProvider):
    """SentenceTransformersProvider (embedding_providers.py:52-80): pooling `cls` | `mean`, L2 normalise."""

    def __init__(self, engine: Any, tokenizer: Any, pooling: str = "cls", normalize: bool = True, max_length: int = 512):
        super().__init__(engine, tokenizer, max_length)
        if pooling not in ("cls", "mean"):
            raise ValueError("pooling must be 'cls' or 'mean'")
        self.pooling, self.normalize = pooling, normalize

    @classmethod
    def from_directory(cls, model_path: str, device: int = 0, max_length: int = 512, pooling: Optional[str] = None,
                       normalize: bool = True, operand_dtype: str = "f16") -> "GpuDenseProvider":
        """`SentenceTransformersProvider(model_name, device)` (embedding_providers.py:55-71) for a checkpoint on disk;
        the pooling mode comes from the checkpoint's `1_Pooling/config.json` unless given."""
        engine, tokenizer, _cfg = load_encoder_directory(model_path, device=device, max_seq_len=max_length, operand_dtype=operand_dtype)
        self = cls(engine, tokenizer, pooling=pooling or _st_pooling_mode(model_path), normalize=normalize, max_length=max_length)
        if operand_dtype == "f16":
            self._rebuild_bf16 = lambda: load_encoder_directory(model_path, device=device, max_seq_len=max_length, operand_dtype="bf16")[0]
        return self

    def _rows(self, texts: Sequence[str]) -> np.ndarray:
        seqs = self._encode(texts)
        out = np.empty((len(seqs), self.engine.shape.hidden_size), np.float32)
        with self._lock:
            for a, b in self._batches(seqs):
                while True:
                    self.engine.load_batch(seqs[a:b])
                    n = b - a
                    ends = [0] * n if self.pooling == "cls" else [len(s) - 1 for s in seqs[a:b]]
                    self.engine.load_ranges(list(range(n)), [0] * n, ends)
                    self.engine.run()
                    self.engine.run_pool(self.normalize)
                    out[a:b] = self.engine.read_pool()
                    if not self._f16_clamped():
                        break
        return out

    def embed_text(self, text: str) -> List[float]:
        return self._rows([text])[0].tolist()

    def embed_batch(self, texts: List[str]) -> List[List[float]]:
        return self._rows(texts).tolist()

    def embed_queries(self, texts: List[str]) -> List[List[float]]:
        """`[embed_text(t) for t in texts]` as shared device batches (rows do not depend on their batch mates)."""
        return self._rows(texts).tolist()

    def get_dimension(self) -> int:
        return int(self.engine.shape.hidden_size)
