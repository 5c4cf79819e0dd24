"""GPU cross-encoder reranker behind the reference's `Reranker` interface.

Kept: `Reranker.rerank(question, results) -> List[SearchResult]` / `rerank_async`, `BaseReranker`'s `rerank_k` /
`text_field` handling and the head/tail split (verbatim_rag/rerankers.py:14-41), the ordering rule of
`SentenceTransformersReranker.rerank` (`sorted(zip(scores, head), reverse=True)`, :128-134); the caller is
`VerbatimRAG._apply_reranker` (verbatim_rag/core.py:125-140).  Replaced: sentence-transformers `CrossEncoder.predict`
(third-party, absent here) by the HIP BERT-family encoder + pooler/classifier head on packed `[CLS] q [SEP] d [SEP]`
pairs with token types 0 / 1 (`cross-encoder/ms-marco-MiniLM-L-6-v2` is a 6-layer, 384-wide, 12-head BERT: head_dim
32, run on the head_dim-64 kernels with zero-padded heads).  Scores are the classifier logits; CrossEncoder applies a
monotone activation (identity or sigmoid) to a single label, which does not change the order.
"""
from __future__ import annotations

import asyncio
import threading
from abc import ABC, abstractmethod
from typing import Any, List, Sequence, Tuple


class Reranker(ABC):
    @abstractmethod
    def rerank(self, question: str, results: List[Any]) -> List[Any]:
        raise NotImplementedError

    async def rerank_async(self, question: str, results: List[Any]) -> List[Any]:
        return await asyncio.to_thread(self.rerank, question, results)


class BaseReranker(Reranker):
    def __init__(self, rerank_k: int = 50, text_field: str = "text"):
        self.rerank_k = rerank_k
        self.text_field = text_field

    def _split_results(self, results: List[Any]):
        return results[: self.rerank_k], results[self.rerank_k:]

    def _get_texts(self, results: List[Any]) -> List[str]:
        if self.text_field == "enhanced_text":
            return [r.enhanced_text or r.text for r in results]
        return [r.text for r in results]


def pack_pair(q_ids: Sequence[int], d_ids: Sequence[int], cls_id: int, sep_id: int, max_length: int) -> Tuple[List[int], List[int]]:
    """`[CLS] q [SEP] d [SEP]` with token types 0 / 1, truncated to `max_length` the way HF fast tokenizers truncate a
    pair with `truncation=True` (strategy `longest_first`, tokenizers `utils/truncation.rs`): with budget =
    max_length - 3 special tokens, the shorter side keeps min(len, budget // 2 when both overflow) and the longer side
    takes the rest; on equal lengths the first sequence counts as the shorter one."""
    n1, n2 = len(q_ids), len(d_ids)
    budget = max_length - 3
    if n1 + n2 > budget:
        swap = n1 > n2
        if swap:
            n1, n2 = n2, n1
        n2 = n1 if n1 > budget else max(n1, budget - n1)
        if n1 + n2 > budget:
            n1 = budget // 2
            n2 = n1 + budget % 2
        if swap:
            n1, n2 = n2, n1
    q, d = list(q_ids[:n1]), list(d_ids[:n2])
    ids = [cls_id] + q + [sep_id] + d + [sep_id]
    types = [0] * (len(q) + 2) + [1] * (len(d) + 1)
    return ids, types


class GpuCrossEncoderReranker(BaseReranker):
    """SentenceTransformersReranker (verbatim_rag/rerankers.py:109-134) on a `BertEncoderEngine` with a pair head."""

    def __init__(self, engine: Any, tokenizer: Any, rerank_k: int = 50, text_field: str = "text", max_length: int = 512):
        super().__init__(rerank_k=rerank_k, text_field=text_field)
        if not getattr(engine, "pair_labels", 0):
            raise ValueError("engine has no pair head (BertForSequenceClassification weights)")
        self.engine, self.tokenizer = engine, tokenizer
        self.max_length = min(max_length, engine.max_seq_len)
        self._lock = getattr(engine, "lock", None) or threading.Lock()   # the handle's own lock: wrappers may share it

    @classmethod
    def from_directory(cls, model_path: str, device: int = 0, rerank_k: int = 50, max_length: int = 512, **kw) -> "GpuCrossEncoderReranker":
        """`SentenceTransformersReranker(model_name)` (rerankers.py:109-134) for a `BertForSequenceClassification`
        checkpoint on disk (e.g. a downloaded `cross-encoder/ms-marco-MiniLM-L-6-v2`)."""
        from .embedding_providers import load_encoder_directory

        engine, tokenizer, _cfg = load_encoder_directory(model_path, device=device, max_seq_len=max_length)
        return cls(engine, tokenizer, rerank_k=rerank_k, max_length=max_length, **kw)

    def _ids(self, text: str) -> List[int]:
        enc = self.tokenizer.encode(text, add_special_tokens=False)
        return list(enc.ids if hasattr(enc, "ids") else enc)

    def score(self, question: str, texts: Sequence[str]) -> List[float]:
        q = self._ids(question)
        sh = self.engine.shape
        packed = [pack_pair(q, self._ids(t), sh.cls_token_id, sh.sep_token_id, self.max_length) for t in texts]
        scores: List[float] = []
        with self._lock:
            start = 0
            while start < len(packed):
                tok, end = 0, start
                while end < len(packed) and end - start < self.engine.max_seqs and tok + len(packed[end][0]) <= self.engine.max_tokens:
                    tok += len(packed[end][0])
                    end += 1
                if end == start:
                    raise ValueError("a single pair exceeds the engine workspace")
                logits = self.engine.pair_logits([p[0] for p in packed[start:end]], [p[1] for p in packed[start:end]])
                scores.extend(float(x) for x in logits[:, 0])
                start = end
        return scores

    def rerank(self, question: str, results: List[Any]) -> List[Any]:
        head, tail = self._split_results(results)
        if not head:
            return results
        scores = self.score(question, self._get_texts(head))
        ranked = [r for _, r in sorted(zip(scores, head), reverse=True)]   # rerankers.py:133
        return ranked + tail
