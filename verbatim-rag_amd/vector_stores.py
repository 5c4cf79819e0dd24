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


# ---------------------------------------------------------------------------- hybrid_search.py
def sanitize_hybrid_weights(hybrid_weights: Dict[str, float]) -> Dict[str, float]:
    """hybrid_search.py:15-45."""
    if not hybrid_weights:
        raise ValueError("hybrid_weights must be a non-empty dict")
    allowed = {"dense", "sparse", "full_text"}
    cleaned: Dict[str, float] = {}
    for method, weight in hybrid_weights.items():
        if method not in allowed:
            logger.warning("Ignoring unsupported hybrid method '%s'", method)
            continue
        if not isinstance(weight, (int, float)) or weight <= 0:
            logger.warning("Ignoring non-positive weight for method '%s': %s", method, weight)
            continue
        cleaned[method] = float(weight)
    if not cleaned:
        raise ValueError("No valid hybrid_weights after validation")
    return cleaned


def normalize_weights(results_by_method: Dict[str, List], weights: Dict[str, float]) -> Dict[str, float]:
    """hybrid_search.py:48-70."""
    avail = {m: weights.get(m, 0.0) for m in results_by_method}
    total = sum(avail.values())
    if total == 0:
        logger.warning("No non-zero weights for available methods; using equal weights for: %s",
                       list(results_by_method.keys()))
        return {k: 1.0 / len(results_by_method) for k in results_by_method}
    return {k: v / total for k, v in avail.items()}


def merge_hybrid_results(results_by_method: Dict[str, List[dict]], top_k: int, weights: Dict[str, float],
                         rrf_k: int = 60, log_label: str = "") -> List[dict]:
    """Weighted reciprocal-rank fusion, hybrid_search.py:73-129: score[id] += w_m / (rrf_k + rank + 1)
    in method insertion order, stable sort descending, `distance = 1 - score`."""
    nw = normalize_weights(results_by_method, weights)
    scores: Dict[Any, float] = {}
    hit_map: Dict[Any, dict] = {}
    for method, results in results_by_method.items():
        w = nw.get(method, 0.0)
        for rank, hit in enumerate(results):
            hid = hit.get("id")
            if not hid:
                continue
            if hid not in scores:
                scores[hid] = 0.0
                hit_map[hid] = hit
            scores[hid] += w * (1.0 / (rrf_k + rank + 1))
    ordered = sorted(scores.keys(), key=lambda i: scores[i], reverse=True)
    merged = []
    for hid in ordered[:top_k]:
        h = hit_map[hid].copy()
        h["distance"] = 1.0 - scores[hid]
        merged.append(h)
    return merged


def rrf_merge_rows(rows_by_method: Dict[str, np.ndarray], top_k: int, weights: Dict[str, float],
                   rrf_k: int = 60) -> Tuple[np.ndarray, np.ndarray]:
    """`merge_hybrid_results` for a whole batch of queries on row numbers: `rows_by_method[m]` is `[Q, L]` ranked rows
    (-1 = no hit, only as a tail) of at most two methods, in the methods' insertion order.  Returns `rows [Q, top_k]`
    (-1 padded) and `distance [Q, top_k]` (float64, `1 - score`).  Same arithmetic in the same order as the per-query
    routine -- `score = 0.0 + w_first / (rrf_k + rank + 1) (+ w_second / ...)` in float64, stable descending sort, so
    equal scores keep first-seen order -- which `tests/test_store_host_logic.py` checks element for element."""
    methods = list(rows_by_method)
    if not 1 <= len(methods) <= 2:
        raise ValueError("rrf_merge_rows merges one or two methods")
    nw = normalize_weights({m: None for m in methods}, weights)
    first = np.asarray(rows_by_method[methods[0]], dtype=np.int64)
    Q, L1 = first.shape
    contrib1 = nw.get(methods[0], 0.0) * (1.0 / (rrf_k + np.arange(L1, dtype=np.float64) + 1))
    score1 = np.broadcast_to(0.0 + contrib1, (Q, L1)).copy()
    if len(methods) == 1:
        cand_rows, cand_score = first, score1
    else:
        second = np.asarray(rows_by_method[methods[1]], dtype=np.int64)
        L2 = second.shape[1]
        contrib2 = nw.get(methods[1], 0.0) * (1.0 / (rrf_k + np.arange(L2, dtype=np.float64) + 1))
        same = (first[:, :, None] == second[:, None, :]) & (first[:, :, None] >= 0)          # [Q, L1, L2]: at most one per row / column
        score1 += (same * contrib2[None, None, :]).sum(axis=2)                             # a + b (one non-zero term) or a + 0.0
        only2 = ~same.any(axis=1)                                                            # second-method rows not seen before
        cand_rows = np.concatenate([first, np.where(only2, second, -1)], axis=1)
        cand_score = np.concatenate([score1, np.broadcast_to(0.0 + contrib2, (Q, L2))], axis=1)
    key = np.where(cand_rows >= 0, -cand_score, np.inf)
    order = np.argsort(key, axis=1, kind="stable")[:, :top_k]
    rows = np.take_along_axis(cand_rows, order, axis=1)
    dist = 1.0 - np.take_along_axis(cand_score, order, axis=1)
    if rows.shape[1] < top_k:
        pad = top_k - rows.shape[1]
        rows = np.pad(rows, ((0, 0), (0, pad)), constant_values=-1)
        dist = np.pad(dist, ((0, 0), (0, pad)))
    return rows, dist


def convert_hits_to_results(hits: List[dict], dynamic_fields: Optional[List[str]] = None) -> List[SearchResult]:
    """hybrid_search.py:132-175."""
    dynamic_fields = dynamic_fields or []
    out: List[SearchResult] = []
    for hit in hits:
        entity = hit.get("entity", {})
        metadata = entity.get("metadata", {}) or {}
        if isinstance(metadata, str):
            try:
                metadata = json.loads(metadata)
            except Exception:
                metadata = {"raw": metadata}
        for f in dynamic_fields:
            val = entity.get(f)
            if val is not None:
                metadata[f] = val
        out.append(SearchResult(id=hit.get("id"), score=hit.get("distance", 0.0), text=entity.get("text", ""),
                                enhanced_text=entity.get("enhanced_text", ""), metadata=metadata))
    return out


# ---------------------------------------------------------------------------- device indexes
class DenseShard:
    """One GPU's slice of the dense corpus (rows appended in order; ids are local row numbers)."""

    def __init__(self, dim: int, capacity: int, dtype: str = "bf16", device: int = 0):
        self._lib = _lib.load()
        _lib.require_gpu()
        self.dim, self.capacity = dim, capacity
        self._h = C.c_void_p()
        _lib.check("vrag_dense_index_create", self._lib.vrag_dense_index_create(
            dim, capacity, 0 if dtype == "bf16" else 1, device, C.byref(self._h)))

    def add(self, rows: np.ndarray) -> None:
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"rows must be [n, {self.dim}]")
        _lib.check("vrag_dense_index_add", self._lib.vrag_dense_index_add(self._h, rows.ctypes.data_as(_FP), rows.shape[0]))

    def __len__(self) -> int:
        return int(self._lib.vrag_dense_index_size(self._h))

    def search(self, queries: np.ndarray, k: int, stream=None) -> Tuple[np.ndarray, np.ndarray]:
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)
        scores = np.empty((q.shape[0], k), np.float32)
        ids = np.empty((q.shape[0], k), np.int64)
        _lib.check("vrag_dense_index_search", self._lib.vrag_dense_index_search(
            self._h, q.ctypes.data_as(_FP), q.shape[0], k, scores.ctypes.data_as(_FP), ids.ctypes.data_as(_LP), stream))
        return scores, ids

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


def json_serialize_safe(obj: Any) -> Any:
    """What the reference's store does to metadata before the insert (vector_stores/utils.py:10-29): datetimes become
    ISO strings, enums their values, dict keys strings, recursively through dicts and lists -- the JSON column's view of
    the caller's objects, which is what searches return and filters compare against."""
    from datetime import datetime
    from enum import Enum

    if isinstance(obj, datetime):
        return obj.isoformat()
    if isinstance(obj, Enum):
        return getattr(obj, "value", str(obj))
    if isinstance(obj, dict):
        return {str(k): json_serialize_safe(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [json_serialize_safe(item) for item in obj]
    return obj


# ---------------------------------------------------------------------------- the store
_FILTER_TOKEN = re.compile(r"""\s*(?:(?P<meta>metadata\[\s*["'](?P<mkey>[^"']+)["']\s*\])|(?P<str>"[^"]*"|'[^']*')|(?P<num>-?\d+(?:\.\d+)?)"""
                           r"""|(?P<op>==|!=|<=|>=|<|>|&&|\|\||[()\[\],])|(?P<word>\w+))""")


def parse_filter(expr: str):
    """Compiles the subset of Milvus boolean expressions the store supports into `predicate(metadata: dict) -> bool`:
    comparisons `field == value`, `field != value`, `field in [v, ...]`, `field < / <= / > / >= value` (numbers against
    numeric metadata, quoted strings against text metadata) (field = `metadata["key"]`, the Local dialect,
    or a bare `key`, the Cloud dialect: index.py:735-739; values = quoted strings or numbers, compared as strings like
    the JSON-path match on string metadata), combined with `and` / `&&`, `or` / `||`, `not` and parentheses.
    Anything else raises ValueError -- a filter is never silently ignored."""
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
            toks.append(("num", m.group("num")))
        elif m.group("op"):
            toks.append(("op", m.group("op")))
        else:
            w = m.group("word")
            toks.append(("op", w.lower()) if w.lower() in ("and", "or", "not", "in") else ("field", w))
    i = 0

    def peek():
        return toks[i] if i < len(toks) else (None, None)

    def take(kind=None, value=None):
        nonlocal i
        k, v = peek()
        if kind == "val" and k == "num":      # a number where a value is expected (compared as text by == / != / in)
            k = "val"
        if k is None or (kind and k != kind) or (value and v != value):
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
            val = take("val")
            if op == "!=":
                return lambda md: str(md.get(key)) != val
            f = lambda md: str(md.get(key)) == val   # noqa: E731
            f.lookup = (key, [val])                  # lets the store answer from a per-key value index
            return f
        if op == "in":
            take("op", "[")
            vals = [take("val")]
            while peek() == ("op", ","):
                take()
                vals.append(take("val"))
            take("op", "]")
            vs = set(vals)
            f = lambda md: str(md.get(key)) in vs    # noqa: E731
            f.lookup = (key, vals)
            return f
        if op in ("<", "<=", ">", ">="):
            import operator

            cmp = {"<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge}[op]
            kind, raw = peek()
            take("val")
            if kind == "num":       # numeric bound: only numeric metadata values can pass (JSON numbers; bools are not)
                bound = float(raw)
                return lambda md: isinstance(md.get(key), (int, float)) and not isinstance(md.get(key), bool) and cmp(md.get(key), bound)
            return lambda md: isinstance(md.get(key), str) and cmp(md.get(key), raw)   # quoted bound: text order
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


class GpuVectorStore(VectorStore):
    """Exact GPU search store with BaseMilvusStore's behaviour (milvus_base.py:90-127,189-459).

    dense = COSINE (rows and queries are L2-normalised here, so IP on the device equals cosine),
    sparse = IP over shared terms.  Rows live on the host until the first query after an insert
    ("flush"), then in HBM.  `filter` supports the comparison subset of Milvus expressions in `parse_filter`
    (the reference itself only builds `metadata["document_id"] == "..."`, index.py:735-739); anything else is
    rejected loudly.  Filters and deletes act before the search like Milvus' (a selective filter still returns its
    best rows): `_search_batch` re-runs short queries on a cached shard of just the passing rows.
    """

    enable_full_text = False

    def __init__(self, dense_dim: Optional[int] = 384, sparse_vocab: Optional[int] = 30522, enable_dense: bool = True,
                 enable_sparse: bool = True, dense_dtype: str = "bf16", device: int = 0):
        self._lib = _lib.load()
        _lib.require_gpu()
        self.enable_dense, self.enable_sparse = enable_dense, enable_sparse
        self.dense_dim, self.sparse_vocab, self.dense_dtype, self.device = dense_dim, sparse_vocab, dense_dtype, device
        self._ids: List[str] = []
        self._texts: List[str] = []
        self._enh: List[str] = []
        self._meta: List[Dict[str, Any]] = []
        self._dense_rows: List[np.ndarray] = []
        self._sparse_rows: List[Dict[int, float]] = []
        self._alive: List[bool] = []
        self._dense: Optional[DenseShard] = None
        self._sparse: Optional[SparseShard] = None
        self._dirty = False
        # Callers arrive from asyncio.to_thread workers (index.py:552-655 under api/): inserts, deletes, the flush and the
        # cache fills are serialised by this lock; searches run outside it on the shard objects they captured, and a shard
        # that has been replaced or evicted is released by its last user (DenseShard / SparseShard.__del__), never closed
        # under a running search.
        self._mu = threading.RLock()
        self._masks: Dict[str, Optional[np.ndarray]] = {}
        self._value_indexes: Dict[str, Dict[str, np.ndarray]] = {}
        self._all_ids_truthy: Optional[bool] = None
        self._documents: Dict[str, Dict[str, Any]] = {}      # document records (add_documents / get_document)
        self._subsets: Dict[Any, Tuple[Any, np.ndarray]] = {}   # (kind, mask bytes) -> (subset shard, global row of each subset row)

    SUBSET_CACHE = 4
    K_LIMIT = 1024   # vrag_*_index_search: lists of up to 64 per device pass, longer ones as exact pages of 64

    # -------------------------------------------------------------- ingest
    def add_vectors(self, ids, dense_vectors, sparse_vectors, texts, enhanced_texts, metadatas):
        if self.enable_dense and (dense_vectors is None or len(dense_vectors) == 0):
            raise ValueError("Dense vectors required but not provided")          # milvus_base.py:101-104
        if self.enable_sparse and (sparse_vectors is None or len(sparse_vectors) == 0):
            raise ValueError("Sparse vectors required but not provided")
        with self._mu:
            self._add_locked(ids, dense_vectors, sparse_vectors, texts, enhanced_texts, metadatas)

    def _add_locked(self, ids, dense_vectors, sparse_vectors, texts, enhanced_texts, metadatas):
        for i in range(len(ids)):
            self._ids.append(ids[i])
            self._texts.append(texts[i])
            self._enh.append(enhanced_texts[i])
            self._meta.append(json_serialize_safe(dict(metadatas[i] or {})))   # milvus_base.py:108-109
            self._alive.append(True)
            if self.enable_dense:
                v = np.asarray(dense_vectors[i], dtype=np.float32)
                n = float(np.sqrt((v * v).sum(dtype=np.float32)))
                self._dense_rows.append(v / n if n > 0 else v)                  # COSINE == IP on unit rows
            if self.enable_sparse:
                self._sparse_rows.append({int(k): float(v) for k, v in sparse_vectors[i].items()})
        self._dirty = True
        self._drop_subsets()
        self._value_indexes.clear()
        self._all_ids_truthy = None

    def delete(self, ids: List[str]):
        kill = set(ids)
        with self._mu:
            for i, x in enumerate(self._ids):
                if x in kill:
                    self._alive[i] = False
            self._drop_subsets()

    def _flush(self):
        with self._mu:
            if not self._dirty:
                return
            n = len(self._ids)
            if self.enable_dense:
                self._dense = None                      # released now unless a search on another thread still holds it
                dense = DenseShard(self.dense_dim, max(n, 1), self.dense_dtype, self.device)
                if n:
                    dense.add(np.stack(self._dense_rows))
                self._dense = dense
            if self.enable_sparse:
                self._sparse = None
                self._sparse = SparseShard(self.sparse_vocab, *dicts_to_csr(self._sparse_rows), device=self.device) if n else None
            self._dirty = False

    def _main_shard(self, kind: str):
        """(shard, rows it holds) after a flush, captured under the lock."""
        with self._mu:
            self._flush()
            return (self._dense if kind == "dense" else self._sparse), len(self._ids)

    # -------------------------------------------------------------- search
    def _mask(self, filter: Optional[str]) -> Optional[np.ndarray]:
        """Rows a query may return (alive and passing `filter`), or None for all; cached per filter string until the
        next insert / delete (one Python predicate call per row otherwise, on every query)."""
        key = filter or ""
        with self._mu:
            return self._mask_locked(key, filter)

    def _mask_locked(self, key: str, filter: Optional[str]) -> Optional[np.ndarray]:
        if key not in self._masks:
            alive = np.asarray(self._alive, dtype=bool)
            if filter:
                pred = parse_filter(filter)
                lookup = getattr(pred, "lookup", None)
                if lookup is not None:      # one `==` / `in` comparison (the reference's document_id filter, index.py:735-739)
                    index = self._value_index(lookup[0])
                    passing = np.zeros(len(self._meta), dtype=bool)
                    for v in lookup[1]:
                        rows = index.get(v)
                        if rows is not None:
                            passing[rows] = True
                    alive = alive & passing
                else:
                    alive = alive & np.asarray([bool(pred(md)) for md in self._meta], dtype=bool)
            if len(self._masks) >= 64:
                self._masks.clear()
            self._masks[key] = None if alive.all() else alive
        return self._masks[key]

    def _value_index(self, key: str) -> Dict[str, np.ndarray]:
        """str(metadata[key]) -> rows, built once per key until the next insert: a per-document filter then costs its
        matches, not a Python predicate call per stored row."""
        with self._mu:
            index = self._value_indexes.get(key)
            if index is None:
                buckets: Dict[str, List[int]] = {}
                for i, md in enumerate(self._meta):
                    buckets.setdefault(str(md.get(key)), []).append(i)
                index = self._value_indexes[key] = {v: np.asarray(rows, dtype=np.int64) for v, rows in buckets.items()}
            return index

    def _hit(self, row: int, score: float) -> dict:
        """A search hit in the shape `merge_hybrid_results` works on; the entity (text, metadata copy) is attached
        by `_results` only to the hits that survive the merge."""
        return {"id": self._ids[row], "distance": float(score), "_row": row}

    def _results(self, hits: List[dict]) -> List[SearchResult]:
        full = []
        for h in hits:
            row = h["_row"]
            full.append({"id": h["id"], "distance": h["distance"],
                         "entity": {"text": self._texts[row], "enhanced_text": self._enh[row], "metadata": dict(self._meta[row])}})
        return convert_hits_to_results(full)

    def _search(self, kind: str, query, limit: int, mask: Optional[np.ndarray]) -> List[dict]:
        return self._search_batch(kind, [query], limit, mask)[0]

    def _device_topk(self, kind: str, shard, queries: Sequence[Any], k: int):
        if kind == "dense":   # COSINE: unit queries against the unit rows (fp32 norm, one row at a time or all at once: same bits)
            rows_q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32).reshape(len(queries), self.dense_dim))
            norms = np.sqrt((rows_q * rows_q).sum(axis=1, dtype=np.float32))
            return shard.search(rows_q / np.where(norms > 0, norms, np.float32(1.0))[:, None], k)
        return shard.search(queries, k)       # dicts_to_csr converts keys / weights to int32 / float32

    def _subset(self, kind: str, mask: np.ndarray):
        """A shard holding only the rows that pass `mask` (Milvus filters before it searches, milvus_base.py:240-262, so
        a selective filter must still return its best rows however far down the unfiltered ranking they are).  Built
        from the host copies, cached per (kind, mask) until the next insert / delete; subset row j is global row idx[j],
        idx ascending, so the kernels' `(score desc, id asc)` order carries over."""
        key = (kind, mask.tobytes())
        with self._mu:
            hit = self._subsets.get(key)
            if hit is None:
                idx = np.nonzero(mask)[0]
                if kind == "dense":
                    shard = DenseShard(self.dense_dim, len(idx), self.dense_dtype, self.device)
                    shard.add(np.stack([self._dense_rows[i] for i in idx]))
                else:
                    shard = SparseShard(self.sparse_vocab, *dicts_to_csr([self._sparse_rows[i] for i in idx]), device=self.device)
                while len(self._subsets) >= self.SUBSET_CACHE:
                    self._subsets.pop(next(iter(self._subsets)))          # freed when its last user lets go
                hit = self._subsets[key] = (shard, idx)
            return hit

    def _drop_subsets(self):
        with self._mu:
            self._subsets.clear()
            self._masks.clear()

    def _topk_rows(self, kind: str, queries: Sequence[Any], limit: int, mask: Optional[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
        """Best `limit` (<= 1024) rows per query among the rows that pass `mask`: `rows [Q, limit]` (-1 = no hit, tail
        only) and their fp32 scores.  One device pass for the whole batch over the full shard; queries that come up
        short because filtered / deleted rows took their slots (and every query when the filter passes under 1/8 of
        the rows) get a second pass over the masked subset shard."""
        main, n = self._main_shard(kind)
        if mask is not None and len(mask) != n:     # rows were inserted after the caller built its mask
            mask = np.concatenate([mask, np.zeros(n - len(mask), dtype=bool)]) if len(mask) < n else mask[:n]
        Q = len(queries)
        rows_out = np.full((Q, limit), -1, np.int64)
        score_out = np.zeros((Q, limit), np.float32)
        k = min(self.K_LIMIT, limit)
        n_pass = n if mask is None else int(mask.sum())
        if n == 0 or Q == 0 or n_pass == 0 or main is None:
            return rows_out, score_out
        want = min(k, n_pass)
        short = np.ones(Q, dtype=bool)
        if mask is None or n_pass * 8 >= n:
            scores, rows = self._device_topk(kind, main, queries, k)
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
            score_out[done, :k] = scores_c[done]
            short = ~done
        if short.any():
            shard, idx = self._subset(kind, mask)
            which = np.nonzero(short)[0]
            scores, rows = self._device_topk(kind, shard, [queries[i] for i in which], want)
            found = rows >= 0
            rows_out[which, :want] = np.where(found, idx[np.where(found, rows, 0)], -1)
            score_out[which, :want] = scores
        return rows_out, score_out

    def _search_batch(self, kind: str, queries: Sequence[Any], limit: int, mask: Optional[np.ndarray]) -> List[List[dict]]:
        rows, scores = self._topk_rows(kind, queries, limit, mask)
        return [[self._hit(int(r), float(v)) for r, v in zip(rows[i], scores[i]) if r >= 0] for i in range(len(queries))]

    def _results_rows(self, rows: np.ndarray, distances: np.ndarray) -> List[SearchResult]:
        return self._results([{"id": self._ids[r], "distance": float(d), "_row": int(r)} for r, d in zip(rows, distances) if r >= 0])

    RRF_VECTOR_MAX = 64   # candidate lists up to this long are merged for the whole batch at once (an [Q, L, L] compare)

    def _hybrid_batch(self, dq, sq, top_k, mask, weights, rrf_k) -> List[List[SearchResult]]:
        """Both methods for all queries, then weighted RRF: vectorised over the batch, or the per-query restatement of
        hybrid_search.py when the lists are long or an id is falsy (`merge_hybrid_results` skips such hits)."""
        limit = top_k * 2
        rows_d, sc_d = self._topk_rows("dense", dq, limit, mask)
        rows_s, sc_s = self._topk_rows("sparse", sq, limit, mask)
        if limit <= self.RRF_VECTOR_MAX and self._ids_truthy():
            rows, dist = rrf_merge_rows({"dense": rows_d, "sparse": rows_s}, top_k, weights, rrf_k)
            return [self._results_rows(rows[i], dist[i]) for i in range(len(dq))]
        out = []
        for i in range(len(dq)):
            rbm = {"dense": [self._hit(int(r), float(v)) for r, v in zip(rows_d[i], sc_d[i]) if r >= 0],
                   "sparse": [self._hit(int(r), float(v)) for r, v in zip(rows_s[i], sc_s[i]) if r >= 0]}
            out.append(self._results(merge_hybrid_results(rbm, top_k, weights, rrf_k)))
        return out

    def _ids_truthy(self) -> bool:
        if self._all_ids_truthy is None:
            self._all_ids_truthy = all(bool(x) for x in self._ids)
        return self._all_ids_truthy

    def query_batch(self, dense_queries: Optional[Sequence[Any]] = None, sparse_queries: Optional[Sequence[Any]] = None,
                    text_queries: Optional[Sequence[Optional[str]]] = None, top_k: int = 5, search_type: str = "hybrid",
                    filter: Optional[str] = None, search_params: Optional[Dict[str, Any]] = None,
                    hybrid_weights: Optional[Dict[str, float]] = None, rrf_k: int = 60) -> List[List[SearchResult]]:
        """Cross-query form of `query`: element i equals `query(dense_queries[i], sparse_queries[i], ...)`,
        with the dense and the sparse searches of all queries done as one batched device pass each (the batched
        kernels order hits by `(score desc, id asc)` like the single-query ones).  Queries that take one of `query`'s
        side branches (no vectors, a missing half in hybrid mode) are answered by `query` itself.  On a bf16 shard a
        batch of >= 8 dense queries runs on the matrix cores with the queries rounded to bf16 (include/vrag_amd.h,
        vrag_dense_index_search), so for queries that are not bf16-exact the scores of a batch can differ from the
        single-query scores in the third digit; an f32 shard has no such difference."""
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
            return [self._results_rows(rows[i, :top_k], scores[i, :top_k]) for i in range(n)]   # one method: its first top_k
        if search_type == "dense" and all(is_set(q) for q in dq):
            mask = self._mask(filter)
            rows, scores = self._topk_rows("dense", dq, top_k, mask)
            return [self._results_rows(rows[i], scores[i]) for i in range(n)]
        if search_type == "sparse" and all(is_set(q) for q in sq):
            mask = self._mask(filter)
            rows, scores = self._topk_rows("sparse", sq, top_k, mask)
            return [self._results_rows(rows[i], scores[i]) for i in range(n)]
        if search_type == "hybrid" and all(is_set(q) for q in dq) and all(is_set(q) for q in sq):
            mask = self._mask(filter)
            try:
                return self._hybrid_batch(dq, sq, top_k, mask, {"dense": 0.5, "sparse": 0.5}, rrf_k)
            except Exception as e:
                logger.warning("Batched hybrid search failed: %s, answering per query", e)
        return [single(i) for i in range(n)]

    def query(self, dense_query=None, sparse_query=None, text_query=None, top_k: int = 5, search_type: str = "hybrid",
              filter: Optional[str] = None, search_params: Optional[Dict[str, Any]] = None,
              hybrid_weights: Optional[Dict[str, float]] = None, rrf_k: int = 60) -> List[SearchResult]:
        """milvus_base.py:189-313."""
        if hybrid_weights is not None:
            return self._hybrid_search_with_weights(dense_query, sparse_query, text_query, top_k, filter, hybrid_weights, rrf_k)
        if not dense_query and not sparse_query:
            return self._filter_only_query(filter, top_k)
        mask = self._mask(filter)
        if search_type == "dense" and dense_query:
            hits = self._search("dense", dense_query, top_k, mask)
        elif search_type == "sparse" and sparse_query:
            hits = self._search("sparse", sparse_query, top_k, mask)
        elif search_type == "hybrid" and dense_query and sparse_query:
            try:
                rbm = {"dense": self._search("dense", dense_query, top_k * 2, mask),
                       "sparse": self._search("sparse", sparse_query, top_k * 2, mask)}
                hits = merge_hybrid_results(rbm, top_k, {"dense": 0.5, "sparse": 0.5}, rrf_k=rrf_k)
            except Exception as e:  # milvus_base.py:296-306
                logger.warning("Hybrid search failed: %s, falling back to dense search", e)
                hits = self._search("dense", dense_query, top_k, mask)
        else:
            raise ValueError(f"Invalid search configuration: type={search_type}, "
                             f"dense={dense_query is not None}, sparse={sparse_query is not None}")
        return self._results(hits)

    def _filter_only_query(self, filter: Optional[str], limit: int) -> List[SearchResult]:
        mask = self._mask(filter)
        rows = [i for i in range(len(self._ids)) if mask is None or mask[i]][:limit]
        return [SearchResult(id=self._ids[i], score=1.0, metadata=dict(self._meta[i]), text=self._texts[i],
                             enhanced_text=self._enh[i]) for i in rows]

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
    def save(self, path: str) -> None:
        """Writes the store to a directory: `vectors.npz` (packed unit dense rows, sparse CSR) and `rows.json`
        (ids, texts, enhanced texts, metadata; deleted rows are dropped).  The reference persists through the
        Milvus-lite database file (milvus_local.py:39-56); this is the GPU store's own on-disk format."""
        import json
        import os

        os.makedirs(path, exist_ok=True)
        keep = [i for i, a in enumerate(self._alive) if a]
        arrays: Dict[str, np.ndarray] = {}
        if self.enable_dense:
            arrays["dense"] = (np.stack([self._dense_rows[i] for i in keep]).astype(np.float32) if keep
                               else np.zeros((0, self.dense_dim or 0), np.float32))
        if self.enable_sparse:
            indptr, indices, values = dicts_to_csr([self._sparse_rows[i] for i in keep])
            arrays.update(sp_indptr=indptr, sp_indices=indices, sp_values=values)
        np.savez(os.path.join(path, "vectors.npz"), **arrays)
        with open(os.path.join(path, "rows.json"), "w", encoding="utf-8") as f:
            json.dump({"format": 1, "dense_dim": self.dense_dim, "sparse_vocab": self.sparse_vocab,
                       "enable_dense": self.enable_dense, "enable_sparse": self.enable_sparse,
                       "dense_dtype": self.dense_dtype, "ids": [self._ids[i] for i in keep],
                       "texts": [self._texts[i] for i in keep], "enhanced_texts": [self._enh[i] for i in keep],
                       "metadatas": [self._meta[i] for i in keep], "documents": list(self._documents.values())},
                      f, ensure_ascii=False)

    @classmethod
    def load(cls, path: str, device: int = 0) -> "GpuVectorStore":
        import json
        import os

        with open(os.path.join(path, "rows.json"), encoding="utf-8") as f:
            rows = json.load(f)
        if rows.get("format") != 1:
            raise ValueError(f"{path}: unknown GpuVectorStore format {rows.get('format')!r}")
        st = cls(dense_dim=rows["dense_dim"], sparse_vocab=rows["sparse_vocab"], enable_dense=rows["enable_dense"],
                 enable_sparse=rows["enable_sparse"], dense_dtype=rows["dense_dtype"], device=device)
        z = np.load(os.path.join(path, "vectors.npz"))
        n = len(rows["ids"])
        st._ids, st._texts, st._enh = list(rows["ids"]), list(rows["texts"]), list(rows["enhanced_texts"])
        st._meta = [dict(m) for m in rows["metadatas"]]
        st._documents = {d.get("id", ""): dict(d) for d in rows.get("documents", [])}
        st._alive = [True] * n
        if st.enable_dense:
            d = z["dense"]
            if d.shape[0] != n:
                raise ValueError(f"{path}: {d.shape[0]} dense rows for {n} ids")
            st._dense_rows = [d[i] for i in range(n)]
        if st.enable_sparse:
            ip, ix, vv = z["sp_indptr"], z["sp_indices"], z["sp_values"]
            if len(ip) != n + 1:
                raise ValueError(f"{path}: sparse indptr has {len(ip)} entries for {n} ids")
            st._sparse_rows = [{int(k): float(v) for k, v in zip(ix[ip[i]:ip[i + 1]], vv[ip[i]:ip[i + 1]])} for i in range(n)]
        st._dirty = n > 0
        return st
