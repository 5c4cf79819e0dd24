"""GPU vector store behind the reference's `VectorStore` interface.

Kept from the reference (same names, argument meaning, error behaviour):
  * `SearchResult` / `VectorStore`            verbatim_rag/vector_stores/base.py:10-74
  * `BaseMilvusStore.add_vectors/.query` semantics (dense = COSINE, sparse = IP, hybrid = top-2k per
    method then weighted RRF, dense fallback on failure)      vector_stores/milvus_base.py:90-127,189-313,366-459
  * weighted RRF + hit conversion              vector_stores/hybrid_search.py:15-175 (float64 Python semantics)
The Milvus client is replaced by exact brute-force top-k on the GPU (include/vrag_amd.h,
vrag_dense_index_* / vrag_sparse_index_*).  Multi-GPU: rows shard across ranks and per-shard top-k lists are
all-gathered and merged in distributed.py (ShardedTopK).
"""
from __future__ import annotations

import ctypes as C
import json
import logging
import re
import threading
from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

logger = logging.getLogger(__name__)

_FP = C.POINTER(C.c_float)
_LP = C.POINTER(C.c_int64)
_IP = C.POINTER(C.c_int32)


@dataclass
class SearchResult:
    """base.py:10-39."""

    id: str
    score: float
    metadata: Dict[str, Any]
    text: str
    enhanced_text: str = ""

    def __gt__(self, other):
        return self.score > other.score

    def __lt__(self, other):
        return self.score < other.score

    def __eq__(self, other):
        return self.score == other.score

    def __hash__(self):
        return hash((self.id, self.score, self.text, self.enhanced_text))


class VectorStore(ABC):
    """base.py:42-74."""

    @abstractmethod
    def add_vectors(self, ids, dense_vectors, sparse_vectors, texts, enhanced_texts, metadatas):
        pass

    @abstractmethod
    def query(self, dense_query=None, sparse_query=None, text_query=None, top_k: int = 5,
              search_type: str = "hybrid", filter: Optional[str] = None) -> List[SearchResult]:
        pass

    @abstractmethod
    def delete(self, ids: List[str]):
        pass


# ---------------------------------------------------------------------------- rank fusion
# The reference fuses per-method hit lists with weighted reciprocal-rank fusion (vector_stores/hybrid_search.py:15-175).
# Here ONE array routine (`rrf_merge_rows`) does the arithmetic for a whole batch of queries on row numbers; the
# per-query, dict-shaped entry points the reference's callers know are thin adapters over it.  The arithmetic contract
# (pinned by tests/golden/host_fixtures.json, generated from the imported reference): float64,
# `score(id) = sum over methods in insertion order of share(method) / (rrf_k + rank + 1)` with 0-based ranks, ids ordered
# by score descending with ties in first-seen order, reported `distance = 1 - score`.
FUSION_METHODS = ("dense", "sparse", "full_text")


def sanitize_hybrid_weights(hybrid_weights: Dict[str, float]) -> Dict[str, float]:
    """Keeps the entries that name a fusion method with a positive numeric weight, as floats (hybrid_search.py:15-45).
    Raises ValueError for an empty mapping or when nothing survives; dropped entries are logged."""
    if not hybrid_weights:
        raise ValueError("hybrid_weights must be a non-empty dict")
    kept: Dict[str, float] = {}
    for name, w in hybrid_weights.items():
        usable = name in FUSION_METHODS and isinstance(w, (int, float)) and w > 0
        if usable:
            kept[name] = float(w)
        else:
            logger.warning("hybrid_weights: dropping %r: %r (unknown method or non-positive weight)", name, w)
    if not kept:
        raise ValueError("No valid hybrid_weights after validation")
    return kept


def normalize_weights(results_by_method: Dict[str, Any], weights: Dict[str, float]) -> Dict[str, float]:
    """Share of each method that actually produced a list (hybrid_search.py:48-70): its weight over the sum of the
    present methods' weights; equal shares when that sum is zero."""
    present = list(results_by_method)
    mass = [weights.get(m, 0.0) for m in present]
    total = sum(mass)
    if total == 0:
        logger.warning("hybrid search: no weight on any of %s, fusing with equal shares", present)
        return dict.fromkeys(present, 1.0 / len(present))
    return {m: w / total for m, w in zip(present, mass)}


def rrf_merge_rows(rows_by_method: Dict[str, np.ndarray], top_k: int, weights: Dict[str, float],
                   rrf_k: int = 60) -> Tuple[np.ndarray, np.ndarray]:
    """Weighted RRF for a whole batch of queries on row numbers.  `rows_by_method[m]` is `[Q, L_m]` ranked rows, methods
    in insertion order; a negative entry is a rank without a candidate (no hit, or a hit the caller must skip: it still
    occupies its rank).  Returns `rows [Q, top_k]` (-1 padded) and `distance [Q, top_k]` (float64, `1 - score`).

    All lists are laid side by side as one `[Q, L]` candidate matrix and the (query, row) pairs are grouped by ONE stable
    sort of the flattened matrix -- O(Q L log(Q L)) time, O(Q L) memory, whatever the list length.  A group's score
    is accumulated method by method in insertion order (`np.add.at` applies repeated indices one after the other), so
    every float64 sum has the operands and the order of the reference's sequential accumulation; the final order is
    score descending with ties in first-seen order (a lexsort on the group's first position)."""
    methods = list(rows_by_method)
    if not methods:
        raise ValueError("rrf_merge_rows needs at least one method")
    share = normalize_weights(dict.fromkeys(methods), weights)
    lists = [np.asarray(rows_by_method[m], dtype=np.int64) for m in methods]
    Q = lists[0].shape[0]
    cand = np.concatenate(lists, axis=1)                                          # [Q, L]
    L = cand.shape[1]
    rows_out = np.full((Q, top_k), -1, np.int64)
    dist = np.zeros((Q, top_k), np.float64)
    flat = cand.reshape(-1)
    live = np.nonzero(flat >= 0)[0]                                               # flat positions, ascending = (query, position)
    if live.size == 0:
        return rows_out, dist
    qi = live // L
    span = int(flat.max()) + 1
    key = qi * span + flat[live]                                                  # one integer per (query, row) pair
    order = np.argsort(key, kind="stable")                                        # equal pairs stay in position order
    sk = key[order]
    head = np.concatenate([[True], sk[1:] != sk[:-1]])
    gid_sorted = np.cumsum(head) - 1
    n_groups = int(gid_sorted[-1]) + 1
    gid = np.empty(live.size, np.int64)
    gid[order] = gid_sorted
    first_pos = live[order][head]                                                 # flat position of a group's first occurrence
    score = np.zeros(n_groups, np.float64)
    pos_in_q = live - qi * L
    lo = 0
    for m, rows in zip(methods, lists):
        n = rows.shape[1]
        sel = (pos_in_q >= lo) & (pos_in_q < lo + n)
        gain = share.get(m, 0.0) * (1.0 / (rrf_k + np.arange(n, dtype=np.float64) + 1))
        np.add.at(score, gid[sel], gain[pos_in_q[sel] - lo])
        lo += n
    gq = first_pos // L
    final = np.lexsort((first_pos, -score, gq))                                   # query, then score desc, then first seen
    gq_f = gq[final]
    start = np.searchsorted(gq_f, np.arange(Q), side="left")
    rank_in_q = np.arange(n_groups) - start[gq_f]
    keep = rank_in_q < top_k
    rows_out[gq_f[keep], rank_in_q[keep]] = flat[first_pos[final][keep]]
    dist[gq_f[keep], rank_in_q[keep]] = 1.0 - score[final][keep]
    return rows_out, dist


def merge_hybrid_results(results_by_method: Dict[str, List[dict]], top_k: int, weights: Dict[str, float],
                         rrf_k: int = 60, log_label: str = "") -> List[dict]:
    """The per-query, dict-shaped form (hybrid_search.py:73-129): hits are dicts with an "id"; hits whose id is falsy
    are skipped but keep their rank.  Ids are numbered in first-seen order and the batch routine does the rest; each
    merged entry is a copy of the first hit seen for its id with `distance = 1 - score`."""
    number: Dict[Any, int] = {}
    exemplar: List[dict] = []
    coded: Dict[str, np.ndarray] = {}
    for method, hits in results_by_method.items():
        row = np.full((1, len(hits)), -1, np.int64)
        for pos, hit in enumerate(hits):
            key = hit.get("id")
            if not key:
                continue
            if key not in number:
                number[key] = len(exemplar)
                exemplar.append(hit)
            row[0, pos] = number[key]
        coded[method] = row
    if log_label:
        logger.info("hybrid merge (%s): methods=%s rrf_k=%s top_k=%s", log_label, list(coded), rrf_k, top_k)
    rows, dist = rrf_merge_rows(coded, top_k, weights, rrf_k)
    merged = []
    for code, d in zip(rows[0], dist[0]):
        if code >= 0:
            hit = dict(exemplar[int(code)])
            hit["distance"] = float(d)
            merged.append(hit)
    return merged


def _metadata_of(entity: dict) -> Dict[str, Any]:
    raw = entity.get("metadata", {}) or {}
    if not isinstance(raw, str):
        return raw
    try:                                # a JSON column read back as text
        return json.loads(raw)
    except Exception:
        return {"raw": raw}


def convert_hits_to_results(hits: List[dict], dynamic_fields: Optional[List[str]] = None) -> List[SearchResult]:
    """Search hits (`{"id", "distance", "entity": {...}}`, the shape a Milvus client returns and the shape
    `merge_hybrid_results` passes through) -> `SearchResult`s; `dynamic_fields` present on the entity are copied into
    the metadata (hybrid_search.py:132-175)."""
    results = []
    for hit in hits:
        entity = hit.get("entity", {})
        metadata = _metadata_of(entity)
        metadata.update({f: entity[f] for f in (dynamic_fields or ()) if entity.get(f) is not None})
        results.append(SearchResult(id=hit.get("id"), score=hit.get("distance", 0.0), metadata=metadata,
                                    text=entity.get("text", ""), enhanced_text=entity.get("enhanced_text", "")))
    return results


# ---------------------------------------------------------------------------- device indexes
class DenseShard:
    """One GPU's slice of the dense corpus (rows appended in order; ids are local row numbers)."""

    def __init__(self, dim: int, capacity: int, dtype: str = "bf16", device: int = 0, prefilter: bool = True):
        """dtype "bf16" | "f32".  fp32 rows keep a bf16 prefilter image beside them unless `prefilter=False` (+50 % memory):
        a search ranks the image for 64 candidates per query and re-scores those exactly -- same bits as the full fp32 scan
        (include/vrag_amd.h, dtype 2), half the time for one query and a quarter for a batch of 256."""
        self._lib = _lib.load()
        _lib.require_gpu()
        self.dim, self.capacity = dim, capacity
        self._h = C.c_void_p()
        code = 0 if dtype == "bf16" else (2 if prefilter and dim % 4 == 0 else 1)
        _lib.check("vrag_dense_index_create", self._lib.vrag_dense_index_create(dim, capacity, code, device, C.byref(self._h)))

    def add(self, rows: np.ndarray) -> None:
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"rows must be [n, {self.dim}]")
        _lib.check("vrag_dense_index_add", self._lib.vrag_dense_index_add(self._h, rows.ctypes.data_as(_FP), rows.shape[0]))

    def add_device(self, ptr: int, n: int, stream=None) -> None:
        """Rows already in HBM (`ptr`: device address of fp32 `[n, dim]` on the shard's device): no host round trip."""
        _lib.check("vrag_dense_index_add_device", self._lib.vrag_dense_index_add_device(self._h, C.c_void_p(int(ptr)), int(n), stream))

    def __len__(self) -> int:
        return int(self._lib.vrag_dense_index_size(self._h))

    def search(self, queries: np.ndarray, k: int, stream=None) -> Tuple[np.ndarray, np.ndarray]:
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)
        scores = np.empty((q.shape[0], k), np.float32)
        ids = np.empty((q.shape[0], k), np.int64)
        _lib.check("vrag_dense_index_search", self._lib.vrag_dense_index_search(
            self._h, q.ctypes.data_as(_FP), q.shape[0], k, scores.ctypes.data_as(_FP), ids.ctypes.data_as(_LP), stream))
        return scores, ids

    def search_device(self, queries: np.ndarray, k: int, out_scores: int, out_ids: int, row_map: Optional[int] = None,
                      n_map: int = 0, id_base: int = 0, stream=None) -> None:
        """The same search with the `[Q, k]` lists left in HBM at the device addresses `out_scores` / `out_ids`
        (global ids through the device table `row_map`, or `id_base + row`); kernels are only enqueued on `stream`."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)
        _lib.check("vrag_dense_index_search_device", self._lib.vrag_dense_index_search_device(
            self._h, q.ctypes.data_as(_FP), q.shape[0], k, C.c_void_p(row_map) if row_map else None, n_map, id_base,
            C.c_void_p(out_scores), C.c_void_p(out_ids), stream))

    def run_resident(self, nq: int, k: int, stream=None) -> None:
        _lib.check("vrag_dense_index_run_resident", self._lib.vrag_dense_index_run_resident(self._h, nq, k, stream))

    def close(self):
        if self._h:
            self._lib.vrag_dense_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dicts_to_csr(rows: Sequence[Dict[int, float]]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """`{term: weight}` rows -> CSR (int64 indptr, int32 terms ascending within a row, float32 weights).  One pass of
    C-level iteration plus a lexsort: a 1 M-document ingest or a 1 000-query batch does not loop in Python per entry."""
    from itertools import chain

    n = len(rows)
    lens = np.fromiter((len(r) for r in rows), np.int64, n)
    indptr = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=indptr[1:])
    total = int(indptr[-1])
    terms = np.fromiter(chain.from_iterable(rows), np.int64, total)                       # iterating a dict yields its keys
    weights = np.fromiter(chain.from_iterable(r.values() for r in rows), np.float64, total)
    i32 = np.iinfo(np.int32)
    if total and (terms.min() < i32.min or terms.max() > i32.max):
        # checked on the int64 keys: a term that does not fit int32 would wrap in the cast below and pass the later range checks
        # (terms that fit but lie outside the vocabulary are rejected there, by the caller or the C layer)
        bad = int(np.searchsorted(indptr, np.nonzero((terms < i32.min) | (terms > i32.max))[0][0], side="right") - 1)
        raise ValueError(f"sparse vector {bad} has a term outside the int32 range")
    order = np.lexsort((terms, np.repeat(np.arange(n, dtype=np.int64), lens)))
    return indptr, terms[order].astype(np.int32), weights[order].astype(np.float32)


class SparseShard:
    """One GPU's slice of the SPLADE corpus (immutable SELL-64 image built from CSR)."""

    def __init__(self, vocab: int, indptr: np.ndarray, indices: np.ndarray, values: np.ndarray, device: int = 0):
        self._lib = _lib.load()
        _lib.require_gpu()
        self.vocab = vocab
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        values = np.ascontiguousarray(values, dtype=np.float32)
        self.n_docs = len(indptr) - 1
        self._h = C.c_void_p()
        _lib.check("vrag_sparse_index_create", self._lib.vrag_sparse_index_create(
            vocab, self.n_docs, indptr.ctypes.data_as(_LP), indices.ctypes.data_as(_IP), values.ctypes.data_as(_FP),
            device, C.byref(self._h)))

    def stats(self) -> Dict[str, int]:
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check("vrag_sparse_index_stats", self._lib.vrag_sparse_index_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"n_docs": a.value, "nnz": b.value, "padded_nnz": c.value}

    def search_csr(self, q_indptr, q_indices, q_values, k: int, stream=None) -> Tuple[np.ndarray, np.ndarray]:
        q_indptr = np.ascontiguousarray(q_indptr, dtype=np.int64)
        q_indices = np.ascontiguousarray(q_indices, dtype=np.int32)
        q_values = np.ascontiguousarray(q_values, dtype=np.float32)
        nq = len(q_indptr) - 1
        scores = np.empty((nq, k), np.float32)
        ids = np.empty((nq, k), np.int64)
        _lib.check("vrag_sparse_index_search", self._lib.vrag_sparse_index_search(
            self._h, q_indptr.ctypes.data_as(_LP), q_indices.ctypes.data_as(_IP), q_values.ctypes.data_as(_FP), nq, k,
            scores.ctypes.data_as(_FP), ids.ctypes.data_as(_LP), stream))
        return scores, ids

    def search(self, queries: Sequence[Dict[int, float]], k: int, stream=None):
        return self.search_csr(*dicts_to_csr(queries), k, stream)

    def search_device(self, queries: Sequence[Dict[int, float]], k: int, out_scores: int, out_ids: int,
                      row_map: Optional[int] = None, n_map: int = 0, id_base: int = 0, stream=None) -> None:
        """`search` with the lists left in HBM (see DenseShard.search_device)."""
        q_indptr, q_indices, q_values = dicts_to_csr(queries)
        _lib.check("vrag_sparse_index_search_device", self._lib.vrag_sparse_index_search_device(
            self._h, q_indptr.ctypes.data_as(_LP), q_indices.ctypes.data_as(_IP), q_values.ctypes.data_as(_FP),
            len(q_indptr) - 1, k, C.c_void_p(row_map) if row_map else None, n_map, id_base, C.c_void_p(out_scores),
            C.c_void_p(out_ids), stream))

    def run_resident(self, nq: int, k: int, stream=None) -> None:
        _lib.check("vrag_sparse_index_run_resident", self._lib.vrag_sparse_index_run_resident(self._h, nq, k, stream))

    def close(self):
        if self._h:
            self._lib.vrag_sparse_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_JSON_PLAIN = (str, int, float, bool, type(None))
_NO_METADATA: Dict[str, Any] = {}


def json_serialize_safe(obj: Any) -> Any:
    """The JSON column's view of caller metadata -- what a search returns and what filters compare against
    (the reference applies the same conversion before its insert, vector_stores/utils.py:10-29): datetimes as ISO
    strings, enum members as their values, mapping keys as strings, descending through dicts and lists only."""
    from datetime import datetime
    from enum import Enum

    def view(x):
        if isinstance(x, dict):
            return {str(key): view(val) for key, val in x.items()}
        if isinstance(x, list):
            return [view(item) for item in x]
        if isinstance(x, datetime):
            return x.isoformat()
        if isinstance(x, Enum):
            return getattr(x, "value", str(x))
        return x

    return view(obj)


def _metadata_rows(metadatas: Sequence[Optional[dict]]) -> List[Dict[str, Any]]:
    """`json_serialize_safe(dict(md))` for a batch of rows (milvus_base.py:108-109); flat dicts of plain JSON scalars
    under string keys -- what ingestion produces -- are copied without the recursive walk."""
    plain = _JSON_PLAIN
    out = []
    for md in metadatas:
        if not md:
            out.append(_NO_METADATA)         # stored metadata is never modified in place (results hand out copies)
        elif all(type(k) is str and type(v) in plain for k, v in md.items()):
            out.append(dict(md))
        else:
            out.append(json_serialize_safe(dict(md)))
    return out


class _Column:
    """Append-only numpy column with amortised growth: `data` is the `[n]` or `[n, width]` view of the filled part.
    A view handed out earlier stays valid (it keeps its buffer) and covers the rows that existed then."""

    def __init__(self, dtype, width: Optional[int] = None, first: Optional[Sequence] = None):
        self._width = width
        self._buf = np.empty((0,) if width is None else (0, width), dtype)
        self._n = 0
        if first is not None:
            self.extend(np.asarray(first, dtype))

    def __len__(self) -> int:
        return self._n

    @property
    def data(self) -> np.ndarray:
        return self._buf[: self._n]

    def extend(self, rows: np.ndarray, adopt: bool = False) -> None:
        """`adopt=True`: `rows` is a fresh array nobody else holds -- an empty column takes it as its buffer (no copy)."""
        m = len(rows)
        if adopt and self._n == 0 and rows.dtype == self._buf.dtype and rows.flags.c_contiguous and rows.flags.owndata:
            self._buf, self._n = rows, m
            return
        if self._n + m > len(self._buf):
            cap = max(self._n + m, int(len(self._buf) * 1.5) + 16)
            grown = np.empty((cap,) + self._buf.shape[1:], self._buf.dtype)
            grown[: self._n] = self._buf[: self._n]
            self._buf = grown
        self._buf[self._n: self._n + m] = rows
        self._n += m


def csr_take_rows(indptr: np.ndarray, indices: np.ndarray, values: np.ndarray, rows: np.ndarray):
    """CSR of the selected rows (in the given order), without a Python loop per row."""
    rows = np.asarray(rows, dtype=np.int64)
    lens = indptr[rows + 1] - indptr[rows]
    out_ptr = np.zeros(len(rows) + 1, np.int64)
    np.cumsum(lens, out=out_ptr[1:])
    take = np.repeat(indptr[rows] - out_ptr[:-1], lens) + np.arange(int(out_ptr[-1]), dtype=np.int64)
    return out_ptr, indices[take], values[take]


def _as_csr(sparse_vectors, n_expected: int, vocab: int):
    """Sparse rows in any accepted form -> canonical CSR (int64 indptr, int32 terms strictly ascending within a row,
    float32 weights).  Accepted: the reference's `List[Dict[int, float]]` (embedding_providers.py:33-49), and for bulk
    ingest a `(indptr, indices, values)` triple or a scipy CSR matrix (rows with unsorted terms are sorted; a term
    repeated inside a row is rejected -- a dict cannot hold one)."""
    if hasattr(sparse_vectors, "indptr") and hasattr(sparse_vectors, "indices") and hasattr(sparse_vectors, "data"):
        sparse_vectors = (sparse_vectors.indptr, sparse_vectors.indices, sparse_vectors.data)
    if isinstance(sparse_vectors, tuple) and len(sparse_vectors) == 3 and isinstance(sparse_vectors[0], np.ndarray):
        indptr = np.ascontiguousarray(sparse_vectors[0], dtype=np.int64)
        indices = np.asarray(sparse_vectors[1])
        values = np.ascontiguousarray(sparse_vectors[2], dtype=np.float32)
        if len(indptr) != n_expected + 1:
            raise ValueError(f"add_vectors: {len(indptr) - 1} sparse_vectors for {n_expected} ids")
        if indptr[0] != 0 or (np.diff(indptr) < 0).any() or indptr[-1] != len(indices) or len(indices) != len(values):
            raise ValueError("add_vectors: malformed CSR sparse_vectors")
        if len(indices) and (indices.min() < 0 or indices.max() >= vocab):
            bad = int(np.searchsorted(indptr, np.nonzero((indices < 0) | (indices >= vocab))[0][0], side="right") - 1)
            raise ValueError(f"add_vectors: sparse vector {bad} has a term outside [0, {vocab})")
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        if len(indices) > 1:
            # terms must ascend strictly inside a row: look only at the places where they do not (row starts, mostly)
            drops = np.nonzero(indices[1:] <= indices[:-1])[0] + 1
            if len(drops) and not np.isin(drops, indptr).all():
                row_of = np.repeat(np.arange(n_expected, dtype=np.int64), np.diff(indptr))
                order = np.lexsort((indices, row_of))
                indices, values = indices[order], values[order]
                drops = np.nonzero(indices[1:] <= indices[:-1])[0] + 1
                if not np.isin(drops, indptr).all():
                    raise ValueError("add_vectors: a sparse vector repeats a term")
        return indptr, indices, values
    if len(sparse_vectors) != n_expected:
        raise ValueError(f"add_vectors: {len(sparse_vectors)} sparse_vectors for {n_expected} ids")
    indptr, indices, values = dicts_to_csr(sparse_vectors)
    if len(indices) and (indices.min() < 0 or indices.max() >= vocab):
        bad = int(np.searchsorted(indptr, np.nonzero((indices < 0) | (indices >= vocab))[0][0], side="right") - 1)
        raise ValueError(f"add_vectors: sparse vector {bad} has a term outside [0, {vocab})")
    return indptr, indices, values


def _merge_parts(scores: np.ndarray, rows: np.ndarray, k: int, device: int) -> Tuple[np.ndarray, np.ndarray]:
    """Lists of the segments of one shard `[P, Q, k]` (global rows) -> `[Q, k]`, on the GPU (`vrag_topk_merge`)."""
    from .distributed import merge_topk_device

    return merge_topk_device(scores, rows, k, device)


# ---------------------------------------------------------------------------- filters
_FILTER_TOKEN = re.compile(r"""\s*(?:(?P<meta>metadata\[\s*["'](?P<mkey>[^"']+)["']\s*\])|(?P<str>"[^"]*"|'[^']*')"""
                           r"""|(?P<num>-?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?)"""
                           r"""|(?P<op>==|!=|<=|>=|<|>|&&|\|\||[()\[\],])|(?P<word>\w+))""")


def _typed(value: Any):
    """Comparison key of a metadata value or a filter literal: JSON semantics, not text -- a number equals a number
    (2020 == 2020.0), a string equals a string ("5" != 5), booleans only booleans; anything else (None, a missing
    key, lists, dicts) equals nothing."""
    if isinstance(value, bool):
        return ("b", value)
    if isinstance(value, (int, float)):
        return ("n", float(value))
    if isinstance(value, str):
        return ("s", value)
    return None


def parse_filter(expr: str):
    """Compiles the subset of Milvus boolean expressions the store supports into `predicate(metadata: dict) -> bool`:
    `field == value`, `field != value`, `field in [v, ...]`, `field < / <= / > / >= value`, combined with `and` / `&&`,
    `or` / `||`, `not` and parentheses.  field = `metadata["key"]` (the Local dialect) or a bare `key` (the Cloud
    dialect: index.py:735-739); value = a quoted string, a number (`7`, `-2.5`, `1e3`) or `true` / `false`.
    Comparisons are typed like Milvus' JSON path match: numbers against numeric metadata, strings against text,
    booleans against booleans; a missing key equals nothing (so `!=` holds for it).  Anything else raises ValueError --
    a filter is never silently ignored."""
    toks, pos = [], 0
    while pos < len(expr):
        if expr[pos:].strip() == "":
            break
        m = _FILTER_TOKEN.match(expr, pos)
        if not m:
            raise ValueError(f"GpuVectorStore: cannot parse filter at {expr[pos:]!r}")
        pos = m.end()
        if m.group("meta"):
            toks.append(("field", m.group("mkey")))
        elif m.group("str"):
            toks.append(("val", m.group("str")[1:-1]))
        elif m.group("num"):
            toks.append(("val", float(m.group("num"))))
        elif m.group("op"):
            toks.append(("op", m.group("op")))
        else:
            w = m.group("word")
            if w.lower() in ("and", "or", "not", "in"):
                toks.append(("op", w.lower()))
            elif w.lower() in ("true", "false"):
                toks.append(("val", w.lower() == "true"))
            else:
                toks.append(("field", w))
    i = 0

    def peek():
        return toks[i] if i < len(toks) else (None, None)

    def take(kind=None, value=None):
        nonlocal i
        k, v = peek()
        if k is None or (kind and k != kind) or (value is not None and v != value):
            raise ValueError(f"GpuVectorStore: unsupported filter {expr!r}")
        i += 1
        return v

    def comparison():
        if peek() == ("op", "("):
            take()
            f = disjunction()
            take("op", ")")
            return f
        if peek() == ("op", "not"):
            take()
            g = comparison()
            return lambda md: not g(md)
        key = take("field")
        op = take("op")
        if op in ("==", "!="):
            want = _typed(take("val"))
            if op == "!=":
                return lambda md: _typed(md.get(key)) != want
            f = lambda md: _typed(md.get(key)) == want   # noqa: E731
            f.lookup = (key, [want])                      # lets the store answer from a per-key value index
            return f
        if op == "in":
            take("op", "[")
            vals = [_typed(take("val"))]
            while peek() == ("op", ","):
                take()
                vals.append(_typed(take("val")))
            take("op", "]")
            vs = set(vals)
            f = lambda md: _typed(md.get(key)) in vs      # noqa: E731
            f.lookup = (key, vals)
            return f
        if op in ("<", "<=", ">", ">="):
            import operator

            cmp = {"<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge}[op]
            kind, bound = _typed(take("val"))
            if kind == "b":
                raise ValueError(f"GpuVectorStore: ordering comparison with a boolean in filter {expr!r}")

            def ordered(md):
                have = _typed(md.get(key))
                return have is not None and have[0] == kind and cmp(have[1], bound)
            return ordered
        raise ValueError(f"GpuVectorStore: unsupported operator {op!r} in filter {expr!r}")

    def conjunction():
        f = comparison()
        while peek() in (("op", "and"), ("op", "&&")):
            take()
            g, h = f, comparison()
            f = (lambda a, b: lambda md: a(md) and b(md))(g, h)
        return f

    def disjunction():
        f = conjunction()
        while peek() in (("op", "or"), ("op", "||")):
            take()
            g, h = f, conjunction()
            f = (lambda a, b: lambda md: a(md) or b(md))(g, h)
        return f

    pred = disjunction()
    if i != len(toks):
        raise ValueError(f"GpuVectorStore: unsupported filter {expr!r}")
    return pred


def _replace_into(path: str, name: str, writer) -> None:
    """Writes beside the final name and renames into place: a reader never sees half a file."""
    import os

    tmp = os.path.join(path, f".{name}.tmp{os.getpid()}")
    writer(tmp)
    os.replace(tmp, os.path.join(path, name))


def _write_text(file: str, text: str) -> None:
    with open(file, "w", encoding="utf-8", newline="") as f:
        f.write(text)


def _put_strings(path: str, stem: str, col: Sequence[str]) -> None:
    """A column of strings as `{stem}.txt`: the rows joined by NUL (one C-level join / split for 10^7 rows); a column
    holding a NUL itself or a non-string goes to `{stem}.json` instead."""
    import os

    blob = None
    try:
        blob = "\x00".join(col)
        if blob.count("\x00") != max(0, len(col) - 1):
            blob = None
    except TypeError:
        blob = None
    # new file first (write beside + rename: a reader never sees the column missing), then the other extension's stale file
    if blob is not None:
        _replace_into(path, f"{stem}.txt", lambda tmp: _write_text(tmp, blob))
        stale = f"{stem}.json"
    else:
        _replace_into(path, f"{stem}.json", lambda tmp: _write_text(tmp, json.dumps(list(col), ensure_ascii=False)))
        stale = f"{stem}.txt"
    if os.path.exists(os.path.join(path, stale)):
        os.remove(os.path.join(path, stale))


def _get_strings(path: str, stem: str, n: int) -> List[str]:
    import os

    txt, js = os.path.join(path, f"{stem}.txt"), os.path.join(path, f"{stem}.json")

    def read_txt():
        with open(txt, encoding="utf-8", newline="") as f:
            return f.read().split("\x00") if n else []

    def read_json():
        with open(js, encoding="utf-8") as f:
            return json.load(f)

    if os.path.exists(txt) and os.path.exists(js):
        # an overwrite was interrupted between the rename of the new file and the removal of the old one: the newer file wins;
        # with equal timestamps (coarse clocks, restored backups) the one that holds the manifest's row count does
        mt, mj = os.path.getmtime(txt), os.path.getmtime(js)
        if mt != mj:
            col = read_json() if mj > mt else read_txt()
        else:
            col = read_txt()
            if len(col) != n:
                col = read_json()
    elif os.path.exists(txt):
        col = read_txt()
    else:
        col = read_json()
    if len(col) != n:
        raise ValueError(f"{path}: {stem} holds {len(col)} rows, expected {n}")
    return col


# ---------------------------------------------------------------------------- the store
class GpuVectorStore(VectorStore):
    """Exact GPU search store with BaseMilvusStore's behaviour (milvus_base.py:90-127,189-459).

    dense = COSINE (rows and queries are L2-normalised here, so IP on the device equals cosine),
    sparse = IP over shared terms.  Rows live on the host until the first query after an insert
    ("flush"), then in HBM; a flush appends the new dense rows to the resident shard and builds a SELL-64 image of the
    new sparse rows only (a tail segment beside the main image; the two are folded into one image when the tail
    outgrows a quarter of it).  Dense rows are stored fp32 like the reference's FLOAT_VECTOR field
    (milvus_local.py:109-118): 4 * dim bytes per row in HBM (times `dense_headroom` of append room) and as much on the
    host; `dense_dtype="bf16"` halves the HBM bytes a query streams at the price of rounding the stored rows, which
    also gives up bit-exact ranking between rows whose fp32 scores nearly tie.
    `filter` supports the comparison subset of Milvus expressions in `parse_filter`
    (the reference itself only builds `metadata["document_id"] == "..."`, index.py:735-739); anything else is
    rejected loudly.  Filters and deletes act before the search like Milvus' (a selective filter still returns its
    best rows): `_topk_rows` re-runs short queries on a cached shard of just the passing rows.
    A search may ask for at most `K_LIMIT` = 1024 rows per method (hybrid search asks for 2 * top_k, so top_k <= 512
    there); more raises ValueError -- never a silently shorter list.

    Host layout (columnar, sized for 10^7 rows): ids in one list with a lazily built id -> row table for deletes; unit
    dense rows in one growing `[n, dim]` fp32 array; sparse rows as one growing CSR; liveness as a bool column.
    `add_vectors` also takes an `[n, dim]` ndarray and a CSR triple / scipy CSR matrix for bulk ingest.

    Multi-GPU: `distributed=True` (one process per GPU, torch.distributed initialised, every rank making
    the SAME calls with the SAME arguments) row-shards the store -- each insert batch is cut contiguously over the
    ranks; a rank keeps the vectors, texts and metadata of ITS rows only (`payload="replicated"` keeps texts and
    metadata on every rank instead), ids stay replicated.  A search runs on each rank's shard, ONE all-gather carries
    the per-shard `[Q, k]` lists and every rank merges them on its GPU (`distributed.ShardComm`; under RCCL the lists
    never leave HBM between the local search and the merge); the texts / metadata of the merged hits then meet in one
    object gather.  `query` / `query_batch` return the same results on every rank and the same results as a
    single-GPU store.
    """

    enable_full_text = False

    SUBSET_CACHE = 4
    K_LIMIT = 1024          # vrag_*_index_search: lists of up to 64 per device pass, longer ones as exact pages of 64
    DEVICE_K = 64           # longest list the device-resident exchange carries (one device pass)
    SPARSE_TAIL_MIN = 65536  # rows a sparse tail segment may always hold before it is folded into the main image

    def _use_prefilter(self) -> bool:
        if self.dense_dtype != "f32" or self.dense_prefilter is False:
            return False
        if self.dense_prefilter is True:
            return True
        return self.dense_dim is not None and self.dense_dim % 64 == 0 and self.dense_dim <= 768

    def __init__(self, dense_dim: Optional[int] = 384, sparse_vocab: Optional[int] = 30522, enable_dense: bool = True,
                 enable_sparse: bool = True, dense_dtype: str = "f32", device: int = 0, distributed: bool = False,
                 group=None, comm=None, payload: str = "sharded", dense_headroom: float = 1.5,
                 dense_prefilter="auto"):
        """`dense_prefilter` (fp32 rows only): keep a bf16 image of the rows beside them so that every search streams
        half (small batches) or a fraction (large ones) of the bytes of the full fp32 scan and returns the same bits (`DenseShard`).  It costs
        +50 % of the dense rows' HBM.  "auto" (default) = on where the image route exists (dim % 64 == 0 and <= 768,
        csrc/topk.hip `prefilter_route_ok`), True / False force it; the choice is kept in a saved store's manifest."""
        self._lib = _lib.load()
        _lib.require_gpu()
        if dense_dtype not in ("f32", "bf16"):
            raise ValueError(f"dense_dtype must be 'f32' or 'bf16' (got {dense_dtype!r})")
        if payload not in ("sharded", "replicated"):
            raise ValueError(f"payload must be 'sharded' or 'replicated' (got {payload!r})")
        self.enable_dense, self.enable_sparse = enable_dense, enable_sparse
        self.dense_dim, self.sparse_vocab, self.dense_dtype, self.device = dense_dim, sparse_vocab, dense_dtype, device
        self.dense_headroom = max(1.0, float(dense_headroom))
        if dense_prefilter not in ("auto", True, False):
            raise ValueError(f"dense_prefilter must be 'auto', True or False (got {dense_prefilter!r})")
        self.dense_prefilter = dense_prefilter
        self._comm = comm
        self._owns_comm = False
        if comm is None and distributed:
            from .distributed import ShardComm

            self._comm = ShardComm(group, device)
            self._owns_comm = True
        self._rank = self._comm.rank if self._comm is not None else 0
        self._world = self._comm.world if self._comm is not None else 1
        self._payload_sharded = self._world > 1 and payload == "sharded"
        # replicated on every rank, indexed by global row
        self._ids: List[str] = []
        self._alive = _Column(bool)
        self._id_rows: Optional[Dict[str, Any]] = None        # id -> row (or rows), built by the first delete
        # texts / metadata: indexed by global row, or by local row when the payload is sharded
        self._texts: List[str] = []
        self._enh: List[str] = []
        self._meta: List[Dict[str, Any]] = []
        # this rank's shard, indexed by local row; `_owned.data[j]` = global row of local row j (ascending)
        self._owned = _Column(np.int64)
        self._dense_rows = _Column(np.float32, dense_dim) if enable_dense else None
        self._sp_ptr = _Column(np.int64, first=[0])
        self._sp_idx = _Column(np.int32)
        self._sp_val = _Column(np.float32)
        self._dense: Optional[DenseShard] = None
        self._dense_cap = 0          # capacity of the resident dense shard
        self._dense_flushed = 0      # local rows already in it
        self._sparse_parts: List[Tuple[Any, int, int]] = []   # (SELL image, first local row, rows): main [+ tail]
        self._sparse_flushed = 0
        self._main_rows: np.ndarray = np.zeros(0, np.int64)   # _owned.data at the last flush
        self._owned_dev = None       # the same table in HBM (RCCL exchange): (torch tensor, rows)
        self._dirty = False
        # Callers arrive from asyncio.to_thread workers (index.py:552-655 under api/): inserts, deletes, the flush and the
        # cache fills are serialised by this lock; searches run outside it on the shard objects they captured, and a shard
        # that has been replaced or evicted is released by its last user (DenseShard / SparseShard.__del__), never closed
        # under a running search.  (Sharded stores are SPMD: one caller thread per rank.)
        self._mu = threading.RLock()
        self._masks: Dict[str, Optional[np.ndarray]] = {}
        self._value_indexes: Dict[str, Dict[Any, List[np.ndarray]]] = {}   # key -> typed value -> row segments
        self._all_ids_truthy = True
        self._documents: Dict[str, Dict[str, Any]] = {}      # document records (add_documents / get_document)
        self._subsets: Dict[Any, Tuple[Any, np.ndarray, Any]] = {}   # (kind, mask bytes) -> (subset shard, global row of each subset row, device copy)

    def __len__(self) -> int:
        return len(self._ids)

    def close(self) -> None:
        """Drops the resident shards and, when the store created its own `ShardComm`, the library's RCCL communicator -- call
        it on every rank before `torch.distributed.destroy_process_group()` (a communicator left to the garbage collector may
        be destroyed after torch's process group is gone).  The store is empty afterwards."""
        with self._mu:
            self._subsets.clear()
            self._dense = None              # released by their last user (DenseShard / SparseShard.__del__)
            self._sparse_parts = []
            self._owned_dev = None
            self._dense_flushed = self._sparse_flushed = 0
            self._dirty = True
            if self._comm is not None and self._owns_comm:
                self._comm.close()

    # -------------------------------------------------------------- ingest
    def add_vectors(self, ids, dense_vectors, sparse_vectors, texts, enhanced_texts, metadatas):
        if self.enable_dense and (dense_vectors is None or len(dense_vectors) == 0):
            raise ValueError("Dense vectors required but not provided")          # milvus_base.py:101-104
        if self.enable_sparse and (sparse_vectors is None or
                                   (len(sparse_vectors) == 0 if not hasattr(sparse_vectors, "indptr") else sparse_vectors.shape[0] == 0)):
            raise ValueError("Sparse vectors required but not provided")
        # Build and validate every new row BEFORE touching the store (the reference assembles the whole batch and
        # inserts it in one call, milvus_base.py:90-127): a malformed entry leaves the store exactly as it was.
        n = len(ids)
        for name, col in (("texts", texts), ("enhanced_texts", enhanced_texts), ("metadatas", metadatas)):
            if len(col) != n:
                raise ValueError(f"add_vectors: {len(col)} {name} for {n} ids")
        lo, hi = (0, n)
        if self._world > 1:
            from .distributed import shard_range

            lo, hi = shard_range(n, self._rank, self._world)
        new_dense = None
        if self.enable_dense:
            if len(dense_vectors) != n:
                raise ValueError(f"add_vectors: {len(dense_vectors)} dense_vectors for {n} ids")
            try:
                block = np.asarray(dense_vectors, dtype=np.float32)
            except ValueError:
                block = None                                   # ragged rows: name the first offender below
            if block is None or block.ndim != 2 or block.shape[1] != self.dense_dim:
                for i, v in enumerate(dense_vectors):
                    shape = np.asarray(v, dtype=np.float32).shape
                    if shape != (self.dense_dim,):
                        raise ValueError(f"add_vectors: dense vector {i} has shape {shape}, the store holds {self.dense_dim}-d rows")
                raise ValueError(f"add_vectors: dense_vectors must be [n, {self.dense_dim}]")
            new_dense = np.empty((hi - lo, self.dense_dim), np.float32)
            slab = 16384                                     # a cache-sized slab at a time into one reused scratch: no
            sq = np.empty((min(slab, max(1, hi - lo)), self.dense_dim), np.float32)   # [n, dim] temporaries at 10^6 rows
            for a in range(lo, hi, slab):
                mine = block[a:min(hi, a + slab)]
                np.multiply(mine, mine, out=sq[: len(mine)])
                norms = np.sqrt(sq[: len(mine)].sum(axis=1, dtype=np.float32))
                np.divide(mine, np.where(norms > 0, norms, np.float32(1.0))[:, None], out=new_dense[a - lo:a - lo + len(mine)])   # COSINE == IP on unit rows
        new_csr = None
        if self.enable_sparse:
            indptr, indices, values = _as_csr(sparse_vectors, n, self.sparse_vocab)
            new_csr = (indptr[lo:hi + 1] - indptr[lo], indices[indptr[lo]:indptr[hi]], values[indptr[lo]:indptr[hi]])
        keep = slice(lo, hi) if self._payload_sharded else slice(0, n)
        new_meta = _metadata_rows(metadatas[keep])                                # milvus_base.py:108-109
        new_texts, new_enh = list(texts[keep]), list(enhanced_texts[keep])
        with self._mu:
            base = len(self._ids)
            self._ids.extend(ids)
            if self._id_rows is not None:
                for i, x in enumerate(ids):
                    self._note_id(x, base + i)
            if self._all_ids_truthy:
                self._all_ids_truthy = all(bool(x) for x in ids)
            self._texts.extend(new_texts)
            self._enh.extend(new_enh)
            self._meta.extend(new_meta)
            self._alive.extend(np.ones(n, dtype=bool))
            self._owned.extend(np.arange(base + lo, base + hi, dtype=np.int64))
            if new_dense is not None:
                self._dense_rows.extend(new_dense, adopt=True)
            if new_csr is not None:
                self._sp_ptr.extend(new_csr[0][1:] + self._sp_ptr.data[-1])
                self._sp_idx.extend(new_csr[1])
                self._sp_val.extend(new_csr[2])
            self._dirty = True
            self._drop_subsets()
            # value indexes that exist are extended by the new rows' values (a segment per insert, merged on the next read):
            # rebuilding them is a Python pass over EVERY stored row, per key, after every insert
            meta_base = len(self._meta) - len(new_meta)
            for key, index in self._value_indexes.items():
                fresh: Dict[Any, List[int]] = {}
                for j, md in enumerate(new_meta):
                    t = _typed(md.get(key))
                    if t is not None:
                        fresh.setdefault(t, []).append(meta_base + j)
                for t, rows in fresh.items():
                    index.setdefault(t, []).append(np.asarray(rows, dtype=np.int64))

    def _note_id(self, key, row: int) -> None:
        have = self._id_rows.get(key)
        if have is None:
            self._id_rows[key] = row
        elif isinstance(have, list):
            have.append(row)
        else:
            self._id_rows[key] = [have, row]

    def delete(self, ids: List[str]):
        with self._mu:
            if self._id_rows is None:                       # one pass over the ids, then O(1) per deleted id
                self._id_rows = {}
                for row, key in enumerate(self._ids):
                    self._note_id(key, row)
            alive = self._alive.data
            for key in ids:
                rows = self._id_rows.get(key)
                if rows is not None:
                    alive[rows] = False
            self._drop_subsets()

    def _sparse_slice(self, a: int, b: int):
        """CSR of local rows [a, b)."""
        ptr = self._sp_ptr.data
        return ptr[a:b + 1] - ptr[a], self._sp_idx.data[ptr[a]:ptr[b]], self._sp_val.data[ptr[a]:ptr[b]]

    def _flush(self):
        with self._mu:
            if not self._dirty:
                return
            n = len(self._owned)
            # published before the device append: a search that is running on the resident shard decodes its hits with
            # the mapping it reads AFTER the kernel returns, so every row the kernel can have seen is in it
            self._main_rows = self._owned.data
            if self.enable_dense and n > self._dense_flushed:
                rows = self._dense_rows.data
                if self._dense is None or n > self._dense_cap:
                    # (re)build with head-room so later inserts append instead of re-uploading every row
                    self._dense = None                  # released now unless a search on another thread still holds it
                    self._dense_cap = max(1024, int(n * self.dense_headroom))
                    dense = DenseShard(self.dense_dim, self._dense_cap, self.dense_dtype, self.device, prefilter=self._use_prefilter())
                    dense.add(rows)
                    self._dense = dense
                else:
                    self._dense.add(rows[self._dense_flushed:])
                self._dense_flushed = n
            if self.enable_sparse and n > self._sparse_flushed:
                main = self._sparse_parts[0] if self._sparse_parts else None
                main_n = main[2] if main else 0
                if main is None or n - main_n > max(self.SPARSE_TAIL_MIN, main_n // 4):
                    self._sparse_parts = []             # one image of everything
                    self._sparse_parts = [(SparseShard(self.sparse_vocab, *self._sparse_slice(0, n), device=self.device), 0, n)]
                else:                                   # the main image stays; the rows behind it form the tail segment
                    tail = SparseShard(self.sparse_vocab, *self._sparse_slice(main_n, n), device=self.device)
                    self._sparse_parts = [main, (tail, main_n, n - main_n)]
                self._sparse_flushed = n
            if self._comm is not None and self._comm.on_gpu and n:
                import torch

                self._owned_dev = (torch.from_numpy(np.ascontiguousarray(self._main_rows)).to(torch.device("cuda", self.device)), n)
            self._dirty = False

    def _main_parts(self, kind: str):
        """(segments of the resident shard as (shard, first local row), device row table, rows in the whole store)
        after a flush, captured under the lock."""
        with self._mu:
            self._flush()
            if kind == "dense":
                parts = [(self._dense, 0)] if self._dense is not None else []
            else:
                parts = [(sh, base) for sh, base, _n in self._sparse_parts]
            return parts, self._owned_dev, len(self._ids)

    # -------------------------------------------------------------- row payloads (texts, metadata)
    def _payload_slot(self, row: int) -> int:
        """Index of global row `row` in the text / metadata lists, -1 when another rank holds it."""
        if not self._payload_sharded:
            return row
        owned = self._owned.data
        j = int(np.searchsorted(owned, row))
        return j if j < len(owned) and owned[j] == row else -1

    def _payloads(self, rows) -> Dict[int, Tuple[str, str, Dict[str, Any]]]:
        """global row -> (text, enhanced text, metadata copy) for the given rows; sharded payloads meet in ONE object
        gather (every rank asks for the same rows and contributes the ones it holds)."""
        want = sorted({int(r) for r in rows if r >= 0})
        out = {}
        for r in want:
            j = self._payload_slot(r)
            if j >= 0:
                out[r] = (self._texts[j], self._enh[j], dict(self._meta[j]))
        if self._payload_sharded:
            merged: Dict[int, Tuple[str, str, Dict[str, Any]]] = {}
            for part in self._comm.gather_objects(out):
                merged.update(part)
            return merged
        return out

    # -------------------------------------------------------------- search
    def _mask(self, filter: Optional[str]) -> Optional[np.ndarray]:
        """Rows a query may return (alive and passing `filter`), or None for all; cached per filter string until the
        next insert / delete (one Python predicate call per row otherwise, on every query)."""
        key = filter or ""
        with self._mu:
            return self._mask_locked(key, filter)

    def _mask_locked(self, key: str, filter: Optional[str]) -> Optional[np.ndarray]:
        if key not in self._masks:
            alive = self._alive.data.copy()
            if filter:
                pred = parse_filter(filter)
                lookup = getattr(pred, "lookup", None)
                held = np.zeros(len(self._meta), dtype=bool)     # over the rows whose metadata this rank holds
                if lookup is not None:      # one `==` / `in` comparison (the reference's document_id filter, index.py:735-739)
                    index = self._value_index(lookup[0])
                    for v in lookup[1]:
                        rows = index.get(v)
                        if rows is not None:
                            held[rows] = True
                else:
                    held = np.asarray([bool(pred(md)) for md in self._meta], dtype=bool)
                if self._payload_sharded:
                    passing = np.zeros(len(alive), dtype=bool)
                    passing[self._owned.data[: len(held)][held]] = True
                    passing = self._comm.union_mask(passing)
                else:
                    passing = held
                alive = alive & passing
            if len(self._masks) >= 64:
                self._masks.clear()
            self._masks[key] = None if alive.all() else alive
        return self._masks[key]

    def _value_index(self, key: str) -> Dict[Any, np.ndarray]:
        """typed metadata[key] -> rows (positions in the metadata list), built once per key until the next insert: a
        per-document filter then costs its matches, not a Python predicate call per stored row.  Rows without a
        comparable value are in no bucket."""
        with self._mu:
            index = self._value_indexes.get(key)
            if index is None:
                buckets: Dict[Any, List[int]] = {}
                for i, md in enumerate(self._meta):
                    t = _typed(md.get(key))
                    if t is not None:
                        buckets.setdefault(t, []).append(i)
                index = self._value_indexes[key] = {v: [np.asarray(rows, dtype=np.int64)] for v, rows in buckets.items()}
            out = {}
            for v, segs in index.items():       # buckets are lists of row segments (one per insert since the build): merge on read
                if len(segs) > 1:
                    segs[:] = [np.concatenate(segs)]
                out[v] = segs[0]
            return out

    def _hit(self, row: int, score: float) -> dict:
        """A search hit in the shape `merge_hybrid_results` works on; the entity (text, metadata copy) is attached
        by `_results` only to the hits that survive the merge."""
        return {"id": self._ids[row], "distance": float(score), "_row": row}

    def _results(self, hits: List[dict]) -> List[SearchResult]:
        pay = self._payloads([h["_row"] for h in hits])
        return [SearchResult(id=h["id"], score=h["distance"], metadata=pay[h["_row"]][2], text=pay[h["_row"]][0],
                             enhanced_text=pay[h["_row"]][1]) for h in hits]

    def _search(self, kind: str, query, limit: int, mask: Optional[np.ndarray]) -> List[dict]:
        return self._search_batch(kind, [query], limit, mask)[0]

    def _unit_queries(self, queries: Sequence[Any]) -> np.ndarray:
        """COSINE: unit queries against the unit rows (fp32 norm, one row at a time or all at once: same bits)."""
        rows_q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32).reshape(len(queries), self.dense_dim))
        norms = np.sqrt((rows_q * rows_q).sum(axis=1, dtype=np.float32))
        return rows_q / np.where(norms > 0, norms, np.float32(1.0))[:, None]

    def _device_topk(self, kind: str, parts, shard_rows: Optional[np.ndarray], queries: Sequence[Any], k: int,
                     rows_dev=None):
        """Top-k of `queries` over one (possibly sharded) set of rows -> (`scores [Q, k]`, GLOBAL `rows [Q, k]`, -1 = no
        hit).  `parts` are this rank's segments as (shard, first local row) (empty when it holds none of the rows) and
        `shard_rows[j]` the global row of local row j (None = the resident main shard, whose append-only mapping is read
        after the search; `rows_dev` = the same table in HBM); with more than one rank the per-shard lists meet in one
        all-gather and are merged on the GPU."""
        Q = len(queries)
        comm = self._comm
        q_in = self._unit_queries(queries) if kind == "dense" and parts else queries
        if comm is not None and comm.on_gpu and k <= self.DEVICE_K and (rows_dev is not None or not parts):
            return self._device_topk_resident(parts, rows_dev, q_in, Q, k)
        if not parts:
            scores = np.full((Q, k), -np.inf, np.float32)
            rows = np.full((Q, k), -1, np.int64)
        else:
            found_lists = []
            for shard, base in parts:
                sc, local = shard.search(q_in, k)          # dicts_to_csr converts sparse keys / weights to int32 / float32
                mapping = shard_rows if shard_rows is not None else self._main_rows
                at = local + base
                found = (local >= 0) & (at < len(mapping))
                g = np.where(found, mapping[np.where(found, at, 0)], -1) if len(mapping) else np.full_like(local, -1)
                found_lists.append((sc, g))
            if len(found_lists) == 1:
                scores, rows = found_lists[0]
            else:
                scores, rows = _merge_parts(np.stack([np.where(g >= 0, sc, -np.inf).astype(np.float32) for sc, g in found_lists]),
                                            np.stack([g for _sc, g in found_lists]), k, self.device)
        if comm is not None and (self._world > 1 or comm.on_gpu):
            scores, rows = comm.allgather_merge(scores, rows, k)
        return scores, rows

    def _device_topk_resident(self, parts, rows_dev, q_in, Q: int, k: int):
        """The RCCL form of `_device_topk`: every segment writes its `[Q, k]` lists (global rows through the device
        table) into HBM, segments are merged on the device, the payload goes through ONE all-gather and the cross-rank
        merge; only the merged result is copied to the host."""
        import torch

        comm = self._comm
        stream = C.c_void_p(torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream)
        payload, ids_ptr, scores_ptr = comm.exchange_buffers(Q, k)
        n = Q * k
        if not parts:
            _lib.check("vrag_topk_fill_empty", self._lib.vrag_topk_fill_empty(C.c_void_p(scores_ptr), C.c_void_p(ids_ptr), n, self.device, stream))
        else:
            table, n_map = rows_dev
            if len(parts) == 1:
                slots = [(scores_ptr, ids_ptr)]
            else:
                seg_s = torch.empty((len(parts), n), dtype=torch.float32, device=payload.device)
                seg_i = torch.empty((len(parts), n), dtype=torch.int64, device=payload.device)
                slots = [(seg_s[p].data_ptr(), seg_i[p].data_ptr()) for p in range(len(parts))]
            for (shard, base), (sp, ip) in zip(parts, slots):
                shard.search_device(q_in, k, sp, ip, row_map=table.data_ptr() + 8 * base, n_map=n_map - base, stream=stream)
            if len(parts) > 1:
                _lib.check("vrag_topk_merge", self._lib.vrag_topk_merge(
                    C.c_void_p(seg_s.data_ptr()), C.c_void_p(seg_i.data_ptr()), len(parts), Q, k, k, 0, 0,
                    C.c_void_p(scores_ptr), C.c_void_p(ids_ptr), 1, self.device, stream))
        return comm.allgather_merge_device(payload, Q, k, k)

    def _subset(self, kind: str, mask: np.ndarray):
        """A shard holding only this rank's rows that pass `mask` (Milvus filters before it searches,
        milvus_base.py:240-262, so a selective filter must still return its best rows however far down the unfiltered
        ranking they are).  Built from the host copies, cached per (kind, mask) until the next insert / delete; returns
        (segments, global row of each subset row, device copy of that table) -- rows ascending, so the kernels'
        `(score desc, id asc)` order carries over."""
        key = (kind, mask.tobytes())
        with self._mu:
            hit = self._subsets.get(key)
            if hit is None:
                owned = self._owned.data
                known = owned < len(mask)                     # rows inserted after the caller built its mask are not in it
                local = np.nonzero(known & mask[np.where(known, owned, 0)])[0] if len(owned) else np.zeros(0, np.int64)
                shard = None
                if len(local) and kind == "dense":
                    shard = DenseShard(self.dense_dim, len(local), self.dense_dtype, self.device, prefilter=self._use_prefilter())
                    shard.add(self._dense_rows.data[local])
                elif len(local):
                    shard = SparseShard(self.sparse_vocab, *csr_take_rows(self._sp_ptr.data, self._sp_idx.data, self._sp_val.data, local),
                                        device=self.device)
                rows = owned[local] if len(owned) else local
                dev = None
                if shard is not None and self._comm is not None and self._comm.on_gpu:
                    import torch

                    dev = (torch.from_numpy(np.ascontiguousarray(rows)).to(torch.device("cuda", self.device)), len(rows))
                while len(self._subsets) >= self.SUBSET_CACHE:
                    self._subsets.pop(next(iter(self._subsets)))          # freed when its last user lets go
                hit = self._subsets[key] = ([(shard, 0)] if shard is not None else [], rows, dev)
            return hit

    def _drop_subsets(self):
        with self._mu:
            self._subsets.clear()
            self._masks.clear()

    def _topk_rows(self, kind: str, queries: Sequence[Any], limit: int, mask: Optional[np.ndarray],
                   _retry: int = 0) -> Tuple[np.ndarray, np.ndarray]:
        """Best `limit` rows per query among the rows that pass `mask`: `rows [Q, limit]` (-1 = no hit, tail only) and
        their fp32 scores.  One device pass for the whole batch over the full shard; queries that come up short because
        filtered / deleted rows took their slots (and every query when the filter passes under 1/8 of the rows) get a
        second pass over the masked subset shard.  Every branch below depends only on replicated state and on merged
        results, so the ranks of a sharded store take the same path and meet in the same collectives."""
        if limit > self.K_LIMIT:
            raise ValueError(f"GpuVectorStore: a search may ask for at most {self.K_LIMIT} rows per method "
                             f"(got {limit}; hybrid search asks for 2 * top_k)")
        parts, rows_dev, n = self._main_parts(kind)
        if mask is not None and len(mask) != n:     # rows were inserted after the caller built its mask
            mask = np.concatenate([mask, np.zeros(n - len(mask), dtype=bool)]) if len(mask) < n else mask[:n]
        Q = len(queries)
        rows_out = np.full((Q, limit), -1, np.int64)
        score_out = np.zeros((Q, limit), np.float32)
        k = limit
        n_pass = n if mask is None else int(mask.sum())
        if n == 0 or Q == 0 or n_pass == 0:
            return rows_out, score_out
        want = min(k, n_pass)
        short = np.ones(Q, dtype=bool)
        if mask is None or n_pass * 8 >= n:
            scores, rows = self._device_topk(kind, parts, None, queries, k, rows_dev)
            if (rows >= n).any() and _retry < 3:      # rows inserted while this search ran took slots: search again
                return self._topk_rows(kind, queries, limit, mask, _retry + 1)
            rows = np.where(rows < n, rows, -1)
            found = rows >= 0
            valid = found if mask is None else found & mask[np.where(found, rows, 0)]
            count = valid.sum(axis=1)
            # a sparse query can have fewer than `want` rows sharing a term: then the full pass, which returned fewer
            # than k candidates, has already seen every match
            done = (count >= want) | (found.sum(axis=1) < k) if mask is not None else np.ones(Q, dtype=bool)
            order = np.argsort(~valid, axis=1, kind="stable")                   # passing hits first, ranking kept
            rows_c = np.take_along_axis(rows, order, axis=1)
            scores_c = np.take_along_axis(scores, order, axis=1)
            rows_c[np.arange(k)[None, :] >= count[:, None]] = -1
            rows_out[done, :k] = rows_c[done]
            score_out[done, :k] = np.where(rows_c[done] >= 0, scores_c[done], np.float32(0.0))
            short = ~done
        if short.any():
            sub_parts, shard_rows, sub_dev = self._subset(kind, mask)
            which = np.nonzero(short)[0]
            scores, rows = self._device_topk(kind, sub_parts, shard_rows, [queries[i] for i in which], want, sub_dev)
            rows = np.where((rows >= 0) & (rows < n), rows, -1)
            rows_out[which, :want] = rows
            score_out[which, :want] = np.where(rows >= 0, scores, np.float32(0.0))
        return rows_out, score_out

    def _search_batch(self, kind: str, queries: Sequence[Any], limit: int, mask: Optional[np.ndarray]) -> List[List[dict]]:
        rows, scores = self._topk_rows(kind, queries, limit, mask)
        return [[self._hit(int(r), float(v)) for r, v in zip(rows[i], scores[i]) if r >= 0] for i in range(len(queries))]

    def _results_batch(self, rows: np.ndarray, distances: np.ndarray) -> List[List[SearchResult]]:
        """`[Q, k]` merged rows / scores -> result lists; the payloads of the whole batch are fetched at once."""
        pay = self._payloads(rows.reshape(-1))
        ids = self._ids
        return [[SearchResult(id=ids[r], score=float(d), metadata=dict(pay[r][2]), text=pay[r][0], enhanced_text=pay[r][1])
                 for r, d in zip(rows[i].tolist(), distances[i].tolist()) if r >= 0] for i in range(rows.shape[0])]

    def _hybrid_batch(self, dq, sq, top_k, mask, weights, rrf_k) -> List[List[SearchResult]]:
        """Both methods for all queries, then weighted RRF: one array merge for the whole batch, or query by query when
        an id is falsy (such hits keep their rank but are skipped)."""
        limit = top_k * 2
        rows_d, sc_d = self._topk_rows("dense", dq, limit, mask)
        rows_s, sc_s = self._topk_rows("sparse", sq, limit, mask)
        if self._all_ids_truthy:
            rows, dist = rrf_merge_rows({"dense": rows_d, "sparse": rows_s}, top_k, weights, rrf_k)
            return self._results_batch(rows, dist)
        out = []
        for i in range(len(dq)):
            rbm = {"dense": [self._hit(int(r), float(v)) for r, v in zip(rows_d[i], sc_d[i]) if r >= 0],
                   "sparse": [self._hit(int(r), float(v)) for r, v in zip(rows_s[i], sc_s[i]) if r >= 0]}
            out.append(self._results(merge_hybrid_results(rbm, top_k, weights, rrf_k)))
        return out

    def query_batch(self, dense_queries: Optional[Sequence[Any]] = None, sparse_queries: Optional[Sequence[Any]] = None,
                    text_queries: Optional[Sequence[Optional[str]]] = None, top_k: int = 5, search_type: str = "hybrid",
                    filter: Optional[str] = None, search_params: Optional[Dict[str, Any]] = None,
                    hybrid_weights: Optional[Dict[str, float]] = None, rrf_k: int = 60) -> List[List[SearchResult]]:
        """Cross-query form of `query`: element i equals `query(dense_queries[i], sparse_queries[i], ...)`,
        with the dense and the sparse searches of all queries done as one batched device pass each (the batched
        kernels order hits by `(score desc, id asc)` like the single-query ones).  Queries that take one of `query`'s
        side branches (no vectors, a missing half in hybrid mode) are answered by `query` itself.  On a bf16 shard a
        batch of >= 3 dense queries runs on the matrix cores with every fp32 query carried as a (bf16, bf16 remainder)
        pair (16 significant bits; include/vrag_amd.h, vrag_dense_index_search): scores then agree with the single-query
        fp32 path to fp32 summation noise; an f32 shard (the default) uses fp32 queries at every batch size.
        `top_k` (2 * top_k in hybrid mode) may not exceed `K_LIMIT` = 1024: ValueError otherwise."""
        n = max(len(x) for x in (dense_queries, sparse_queries, text_queries) if x is not None)
        dq = list(dense_queries) if dense_queries is not None else [None] * n
        sq = list(sparse_queries) if sparse_queries is not None else [None] * n
        tq = list(text_queries) if text_queries is not None else [None] * n
        if not (len(dq) == len(sq) == len(tq) == n):
            raise ValueError("query_batch: dense_queries / sparse_queries / text_queries differ in length")

        def single(i):
            return self.query(dense_query=dq[i], sparse_query=sq[i], text_query=tq[i], top_k=top_k, search_type=search_type,
                              filter=filter, search_params=search_params, hybrid_weights=hybrid_weights, rrf_k=rrf_k)

        def is_set(q):   # `query` tests vectors by truthiness (milvus_base.py:232,243,254)
            return q is not None and len(q) > 0

        if hybrid_weights is not None:
            weights = sanitize_hybrid_weights(hybrid_weights)
            if "full_text" in weights and not self.enable_full_text:
                weights = {k: v for k, v in weights.items() if k != "full_text"}
            d_some, s_some = [q is not None for q in dq], [q is not None for q in sq]
            uniform = (all(d_some) or not any(d_some)) and (all(s_some) or not any(s_some))
            use_d, use_s = "dense" in weights and all(d_some), "sparse" in weights and all(s_some)
            if not weights or not uniform or not (use_d or use_s):
                return [single(i) for i in range(n)]          # mixed / degenerate batches: the per-query code decides
            mask = self._mask(filter)
            if use_d and use_s:
                return self._hybrid_batch(dq, sq, top_k, mask, weights, rrf_k)
            rows, scores = self._topk_rows("dense" if use_d else "sparse", dq if use_d else sq, top_k * 2, mask)
            return self._results_batch(rows[:, :top_k], scores[:, :top_k])   # one method: its first top_k
        if search_type == "dense" and all(is_set(q) for q in dq):
            mask = self._mask(filter)
            return self._results_batch(*self._topk_rows("dense", dq, top_k, mask))
        if search_type == "sparse" and all(is_set(q) for q in sq):
            mask = self._mask(filter)
            return self._results_batch(*self._topk_rows("sparse", sq, top_k, mask))
        if search_type == "hybrid" and all(is_set(q) for q in dq) and all(is_set(q) for q in sq):
            mask = self._mask(filter)
            try:
                return self._hybrid_batch(dq, sq, top_k, mask, {"dense": 0.5, "sparse": 0.5}, rrf_k)
            except Exception as e:
                if self._world > 1:
                    raise                                  # ranks must not diverge into different collectives
                logger.warning("Batched hybrid search failed: %s, answering per query", e)
        return [single(i) for i in range(n)]

    def query(self, dense_query=None, sparse_query=None, text_query=None, top_k: int = 5, search_type: str = "hybrid",
              filter: Optional[str] = None, search_params: Optional[Dict[str, Any]] = None,
              hybrid_weights: Optional[Dict[str, float]] = None, rrf_k: int = 60) -> List[SearchResult]:
        """milvus_base.py:189-313.  `top_k` (2 * top_k in hybrid mode) may not exceed `K_LIMIT` = 1024."""
        if hybrid_weights is not None:
            return self._hybrid_search_with_weights(dense_query, sparse_query, text_query, top_k, filter, hybrid_weights, rrf_k)
        if not _is_given(dense_query) and not _is_given(sparse_query):
            return self._filter_only_query(filter, top_k)
        mask = self._mask(filter)
        if search_type == "dense" and _is_given(dense_query):
            hits = self._search("dense", dense_query, top_k, mask)
        elif search_type == "sparse" and _is_given(sparse_query):
            hits = self._search("sparse", sparse_query, top_k, mask)
        elif search_type == "hybrid" and _is_given(dense_query) and _is_given(sparse_query):
            try:
                rbm = {"dense": self._search("dense", dense_query, top_k * 2, mask),
                       "sparse": self._search("sparse", sparse_query, top_k * 2, mask)}
                hits = merge_hybrid_results(rbm, top_k, {"dense": 0.5, "sparse": 0.5}, rrf_k=rrf_k)
            except Exception as e:  # milvus_base.py:296-306
                if self._world > 1:
                    raise
                logger.warning("Hybrid search failed: %s, falling back to dense search", e)
                hits = self._search("dense", dense_query, top_k, mask)
        else:
            raise ValueError(f"Invalid search configuration: type={search_type}, "
                             f"dense={dense_query is not None}, sparse={sparse_query is not None}")
        return self._results(hits)

    def _filter_only_query(self, filter: Optional[str], limit: int) -> List[SearchResult]:
        mask = self._mask(filter)
        rows = (np.arange(min(limit, len(self._ids))) if mask is None else np.nonzero(mask)[0][:limit]).astype(np.int64)
        return self._results_batch(rows[None, :], np.ones((1, len(rows))))[0]

    def _hybrid_search_with_weights(self, dense_query, sparse_query, text_query, top_k, filter, hybrid_weights, rrf_k):
        """milvus_base.py:366-459."""
        hybrid_weights = sanitize_hybrid_weights(hybrid_weights)
        if "full_text" in hybrid_weights and not self.enable_full_text:
            logger.warning("full_text not available on %s, removing from hybrid_weights", self.__class__.__name__)
            hybrid_weights = {k: v for k, v in hybrid_weights.items() if k != "full_text"}
        if not hybrid_weights:
            raise ValueError("No valid search methods in hybrid_weights")
        mask = self._mask(filter)
        rbm: Dict[str, List[dict]] = {}
        if "dense" in hybrid_weights and dense_query is not None:
            rbm["dense"] = self._search("dense", dense_query, top_k * 2, mask)
        if "sparse" in hybrid_weights and sparse_query is not None:
            rbm["sparse"] = self._search("sparse", sparse_query, top_k * 2, mask)
        if len(rbm) == 0:
            logger.warning("Hybrid search: no valid methods executed after validation")
            return []
        if len(rbm) == 1:
            return self._results(list(rbm.values())[0][:top_k])
        return self._results(merge_hybrid_results(rbm, top_k, hybrid_weights, rrf_k))

    def add_documents(self, documents: List[Dict[str, Any]]):
        """Document records beside the chunk rows (milvus_base.py:129-165; probed with `hasattr` by
        VerbatimIndex._store_document_metadata, index.py:299-316): id, title, source, content_type, raw_content,
        metadata, plus the promoted filter fields when the metadata carries them."""
        with self._mu:
            for doc in documents or []:
                metadata = doc.get("metadata", {})
                row = {"id": doc.get("id", ""), "title": doc.get("title") or "", "source": doc.get("source") or "",
                       "content_type": json_serialize_safe(doc.get("doc_type") or doc.get("content_type") or ""),
                       "raw_content": doc.get("raw_content", ""),
                       "metadata": json_serialize_safe(metadata) if isinstance(metadata, dict) else metadata}
                if isinstance(metadata, dict):
                    for key in ("user_id", "dataset_id", "document_id"):
                        if key in metadata:
                            row[key] = metadata.get(key)
                self._documents[row["id"]] = row

    def get_document(self, document_id: str) -> Optional[Dict[str, Any]]:
        """milvus_base.py:173-187: the stored record (a copy) or None."""
        with self._mu:
            row = self._documents.get(document_id)
            return dict(row) if row is not None else None

    # -------------------------------------------------------------- persistence
    FORMAT = 3

    def save(self, path: str) -> None:
        """Writes the store to a directory (deleted rows are dropped): `store.json` (geometry, document records) and the
        id column by rank 0; per rank `vectors.rank{r}.npz` (its packed unit dense rows, sparse CSR and the row numbers
        they belong to) and the text / enhanced-text / metadata columns of those rows (string columns as NUL-joined text,
        `_put_strings`; metadata as one JSON array).  Every file is written beside its final name and renamed into
        place, and a sharded `save` ends in a barrier, so a `load` that follows on any rank reads complete files.  The reference persists through the Milvus-lite database file
        (milvus_local.py:39-56); this is the GPU store's own on-disk format.  Sharded stores: every rank calls `save`
        with the same path."""
        import os

        os.makedirs(path, exist_ok=True)
        with self._mu:
            alive = self._alive.data.copy()
            owned = self._owned.data
            new_row = np.cumsum(alive) - 1                                   # row number after dropping the deleted rows
            local = np.nonzero(alive[owned])[0] if len(owned) else np.zeros(0, np.int64)
            arrays: Dict[str, np.ndarray] = {"owned": new_row[owned[local]].astype(np.int64)}
            if self.enable_dense:
                arrays["dense"] = np.ascontiguousarray(self._dense_rows.data[local], dtype=np.float32)
            if self.enable_sparse:
                indptr, indices, values = csr_take_rows(self._sp_ptr.data, self._sp_idx.data, self._sp_val.data, local)
                arrays.update(sp_indptr=indptr, sp_indices=indices, sp_values=values)
            # texts / metadata lists are indexed by local row, or by global row when a sharded store replicates them
            slots = local if (self._payload_sharded or self._world == 1) else owned[local]
            whole = len(slots) == len(self._texts) and (len(slots) == 0 or (slots[0] == 0 and slots[-1] == len(slots) - 1))

            def take(col, at):        # col[at] for 10^7 rows: one object-array gather instead of a Python loop
                box = np.empty(len(col), dtype=object)
                box[:] = col
                return box[at].tolist()

            pick = (lambda col: col) if whole else (lambda col: take(col, slots))
            texts, enh, metas = pick(self._texts), pick(self._enh), pick(self._meta)
            ids = self._ids if alive.all() else take(self._ids, np.nonzero(alive)[0])
            if not any(metas):
                metas = {"empty_rows": len(metas)}                             # nothing to store per row
            head = {"format": self.FORMAT, "world": self._world, "dense_dim": self.dense_dim, "sparse_vocab": self.sparse_vocab,
                    "enable_dense": self.enable_dense, "enable_sparse": self.enable_sparse, "dense_dtype": self.dense_dtype,
                    "dense_prefilter": self.dense_prefilter, "rows": len(ids), "documents": list(self._documents.values())}
            r = self._rank

            def put_arrays(tmp):
                with open(tmp, "wb") as f:
                    np.savez(f, **arrays)

            _replace_into(path, f"vectors.rank{r}.npz", put_arrays)
            _put_strings(path, f"texts.rank{r}", texts)
            _put_strings(path, f"enhanced.rank{r}", enh)
            _replace_into(path, f"metadatas.rank{r}.json", lambda tmp: _write_text(tmp, json.dumps(metas, ensure_ascii=False)))
            if r == 0:
                _put_strings(path, "ids", ids)
                _replace_into(path, "store.json", lambda tmp: _write_text(tmp, json.dumps(head, ensure_ascii=False)))
        if self._world > 1:
            self._comm.barrier()

    @staticmethod
    def _read_saved(path: str):
        """Any on-disk format -> (head, ids, [per saved rank: (owned rows, arrays, texts, enhanced, metadatas)]).
        Format 3 = `save` above; format 2 = `rows.json` holding ids / texts / metadata for all rows beside
        `vectors.rank{r}.npz`; format 1 = `rows.json` + one `vectors.npz` in row order (single GPU)."""
        import os

        def get_json(name):
            with open(os.path.join(path, name), encoding="utf-8") as f:
                return json.load(f)

        if os.path.exists(os.path.join(path, "store.json")):
            head = get_json("store.json")
            if head.get("format") != 3:
                raise ValueError(f"{path}: unknown GpuVectorStore format {head.get('format')!r}")
            ids = _get_strings(path, "ids", head["rows"])
            shards = []
            for r in range(head.get("world", 1)):
                z = np.load(os.path.join(path, f"vectors.rank{r}.npz"))
                m = len(z["owned"])
                metas = get_json(f"metadatas.rank{r}.json")
                if isinstance(metas, dict):
                    metas = [_NO_METADATA] * int(metas["empty_rows"])
                if len(metas) != m:
                    raise ValueError(f"{path}: metadatas.rank{r} holds {len(metas)} rows, expected {m}")
                shards.append((z["owned"], z, _get_strings(path, f"texts.rank{r}", m), _get_strings(path, f"enhanced.rank{r}", m), metas))
            return head, ids, shards
        rows = get_json("rows.json")
        fmt = rows.get("format")
        if fmt not in (1, 2):
            raise ValueError(f"{path}: unknown GpuVectorStore format {fmt!r}")
        ids = list(rows["ids"])
        head = {k: rows.get(k) for k in ("dense_dim", "sparse_vocab", "enable_dense", "enable_sparse", "dense_dtype")}
        head.update(world=rows.get("world", 1) if fmt == 2 else 1, rows=len(ids), documents=rows.get("documents", []))
        shards = []
        for r in range(head["world"]):
            z = np.load(os.path.join(path, "vectors.npz" if fmt == 1 else f"vectors.rank{r}.npz"))
            owned = np.arange(len(ids), dtype=np.int64) if fmt == 1 else z["owned"]
            shards.append((owned, z, [rows["texts"][g] for g in owned], [rows["enhanced_texts"][g] for g in owned],
                           [rows["metadatas"][g] for g in owned]))
        return head, ids, shards

    @classmethod
    def load(cls, path: str, device: int = 0, distributed: bool = False, group=None, comm=None,
             payload: str = "sharded") -> "GpuVectorStore":
        """Reads a directory written by `save` -- by this or an earlier revision (formats 1 - 3), by any number of
        ranks: when the world size differs from the writer's, the saved shards are put back in row order and cut
        contiguously over the ranks that open the store."""
        head, ids, shards = cls._read_saved(path)
        st = cls(dense_dim=head["dense_dim"], sparse_vocab=head["sparse_vocab"], enable_dense=head["enable_dense"],
                 enable_sparse=head["enable_sparse"], dense_dtype=head["dense_dtype"], device=device,
                 distributed=distributed, group=group, comm=comm, payload=payload,
                 dense_prefilter=head.get("dense_prefilter", "auto"))
        n = len(ids)
        for owned, *_ in shards:
            if len(owned) and (owned.min() < 0 or owned.max() >= n):
                raise ValueError(f"{path}: shard rows outside the {n} stored ids")
        if len(shards) == st._world:
            owned, z, texts, enh, metas = shards[st._rank]
            owned = np.asarray(owned, dtype=np.int64)
            dense = z["dense"] if st.enable_dense else None
            csr = (z["sp_indptr"], z["sp_indices"], z["sp_values"]) if st.enable_sparse else None
        else:                                                     # re-shard: everything back in row order, my contiguous cut
            from .distributed import shard_range

            all_owned = np.concatenate([np.asarray(o, dtype=np.int64) for o, *_ in shards]) if shards else np.zeros(0, np.int64)
            order = np.argsort(all_owned, kind="stable")
            lo, hi = shard_range(n, st._rank, st._world)
            pick = order[lo:hi]                                   # positions in the concatenation of the saved shards
            owned = all_owned[pick]
            texts_all = [t for _o, _z, tx, _e, _m in shards for t in tx]
            enh_all = [t for _o, _z, _tx, e, _m in shards for t in e]
            meta_all = [t for _o, _z, _tx, _e, m in shards for t in m]
            texts, enh, metas = [texts_all[i] for i in pick], [enh_all[i] for i in pick], [meta_all[i] for i in pick]
            dense = np.concatenate([z["dense"] for _o, z, *_ in shards])[pick] if st.enable_dense else None
            csr = None
            if st.enable_sparse:
                ptrs, idxs, vals, off = [np.zeros(1, np.int64)], [], [], 0
                for _o, z, *_ in shards:
                    ptrs.append(z["sp_indptr"][1:].astype(np.int64) + off)
                    idxs.append(z["sp_indices"])
                    vals.append(z["sp_values"])
                    off += int(z["sp_indptr"][-1])
                csr = csr_take_rows(np.concatenate(ptrs), np.concatenate(idxs), np.concatenate(vals), pick)
        m = len(owned)
        covered = np.concatenate([np.asarray(o, dtype=np.int64) for o, *_ in shards]) if shards else np.zeros(0, np.int64)
        if len(covered) != n or len(np.unique(covered)) != n:
            raise ValueError(f"{path}: the saved shards do not cover the {n} stored ids exactly once")
        st._ids = list(ids)
        st._all_ids_truthy = all(bool(x) for x in st._ids)
        st._documents = {d.get("id", ""): dict(d) for d in head.get("documents", [])}
        st._alive.extend(np.ones(n, dtype=bool))
        st._owned.extend(owned)
        if st._payload_sharded or st._world == 1:
            st._texts, st._enh, st._meta = list(texts), list(enh), list(metas)
        else:                                                     # replicated payload: every rank needs every row's
            st._texts, st._enh, st._meta = [""] * n, [""] * n, [{} for _ in range(n)]
            for o, _z, tx, e, mm in shards:
                for g, t1, t2, t3 in zip(np.asarray(o).tolist(), tx, e, mm):
                    st._texts[g], st._enh[g], st._meta[g] = t1, t2, dict(t3)
        if st.enable_dense:
            if dense.shape[0] != m:
                raise ValueError(f"{path}: {dense.shape[0]} dense rows for {m} shard rows")
            st._dense_rows.extend(np.asarray(dense, dtype=np.float32))
        if st.enable_sparse:
            ip, ix, vv = csr
            if len(ip) != m + 1:
                raise ValueError(f"{path}: sparse indptr has {len(ip)} entries for {m} shard rows")
            st._sp_ptr.extend(np.asarray(ip[1:], dtype=np.int64))
            st._sp_idx.extend(np.asarray(ix, dtype=np.int32))
            st._sp_val.extend(np.asarray(vv, dtype=np.float32))
        st._dirty = n > 0
        return st


def _is_given(q) -> bool:
    """Truthiness of a query vector the way the reference tests it (`if dense_query`, milvus_base.py:232-254), for
    lists, dicts and numpy rows alike."""
    return q is not None and len(q) > 0
