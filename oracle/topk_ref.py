"""ctypes loader of oracle/_build/libtopk_ref.so (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtopk_ref.so")
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "topk_ref.c")):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        _LIB = C.CDLL(_SO)
    return _LIB


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 (round to nearest even) -> fp32, the index's storage rounding."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def dense_topk(rows: np.ndarray, queries: np.ndarray, k: int, blocked: bool = False):
    """`blocked=True`: the loop nest for full-size shards (16 queries per pass over a row, rows cut over the OpenMP
    threads); the same fmaf chains, bit-identical results."""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    nq, dim = queries.shape
    scores = np.empty((nq, k), np.float32)
    ids = np.empty((nq, k), np.int64)
    fp, lp = C.POINTER(C.c_float), C.POINTER(C.c_int64)
    fn = _lib().dense_topk_ref_blocked if blocked else _lib().dense_topk_ref
    fn(rows.ctypes.data_as(fp), C.c_int64(rows.shape[0]), C.c_int(dim), queries.ctypes.data_as(fp),
                          C.c_int(nq), C.c_int(k), scores.ctypes.data_as(fp), ids.ctypes.data_as(lp))
    return scores, ids


def sparse_topk(indptr, indices, values, vocab: int, q_indptr, q_indices, q_values, k: int, blocked: bool = False):
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    values = np.ascontiguousarray(values, dtype=np.float32)
    q_indptr = np.ascontiguousarray(q_indptr, dtype=np.int64)
    q_indices = np.ascontiguousarray(q_indices, dtype=np.int32)
    q_values = np.ascontiguousarray(q_values, dtype=np.float32)
    nq = len(q_indptr) - 1
    scores = np.empty((nq, k), np.float32)
    ids = np.empty((nq, k), np.int64)
    fp, lp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    fn = _lib().sparse_topk_ref_blocked if blocked else _lib().sparse_topk_ref
    fn(C.c_int64(len(indptr) - 1), indptr.ctypes.data_as(lp), indices.ctypes.data_as(ip),
                           values.ctypes.data_as(fp), C.c_int(vocab), q_indptr.ctypes.data_as(lp),
                           q_indices.ctypes.data_as(ip), q_values.ctypes.data_as(fp), C.c_int(nq), C.c_int(k),
                           scores.ctypes.data_as(fp), ids.ctypes.data_as(lp))
    return scores, ids
