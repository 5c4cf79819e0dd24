"""Data-parallel sharding of the hot path across the GPUs of one node.

* Extraction / embedding: every (question, chunk) pair is independent -> `shard_range` splits the
  units contiguously over ranks, full weight replica per GPU, NO data-path collective.
* Retrieval: the corpus is row-sharded; queries are replicated (every rank runs the same call, SPMD);
  each rank computes its local top-k on its own GPU; ONE all-gather per query batch carries the
  `[Q, k]` (global row id, fp32 score) lists over RCCL/xGMI (a few KB..MB: latency-bound), then every
  rank merges the world_size sorted lists ON ITS GPU (`vrag_topk_merge`, include/vrag_amd.h) with the
  total order (score desc, id asc).
The reference has no distributed code at all; this is new, not a translation.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional, Tuple

import numpy as np


def shard_range(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split, first `n % world` ranks get one extra unit."""
    q, r = divmod(n_units, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def merge_topk(scores: np.ndarray, ids: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Host statement of the merge (tests and CPU-only tooling; the product merges on the GPU):
    scores/ids `[W, Q, k_in]` per-shard lists (id -1 = empty) -> `[Q, k]` by (score desc, id asc).  One lexsort for the
    whole batch, no per-query Python loop."""
    W, Q, kk = scores.shape
    s = np.ascontiguousarray(scores.transpose(1, 0, 2)).reshape(Q, W * kk).astype(np.float32)
    i = np.ascontiguousarray(ids.transpose(1, 0, 2)).reshape(Q, W * kk).astype(np.int64)
    empty = i < 0
    sort_s = np.where(empty, -np.inf, s).astype(np.float64)
    sort_i = np.where(empty, np.iinfo(np.int64).max, i)
    order = np.lexsort((sort_i, -sort_s), axis=1)[:, :k]      # last key is primary: score desc, then id asc; empties last
    out_s = np.take_along_axis(np.where(empty, -np.inf, s).astype(np.float32), order, axis=1)
    out_i = np.take_along_axis(np.where(empty, -1, i), order, axis=1)
    if out_s.shape[1] < k:
        pad = k - out_s.shape[1]
        out_s = np.pad(out_s, ((0, 0), (0, pad)), constant_values=-np.inf)
        out_i = np.pad(out_i, ((0, 0), (0, pad)), constant_values=-1)
    return out_s, out_i


def merge_topk_device(scores: np.ndarray, ids: np.ndarray, k: int, device: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """`merge_topk` on the GPU from host arrays (`vrag_topk_merge`, on_device = 0)."""
    from . import _lib

    lib = _lib.load()
    W, Q, kk = scores.shape
    s = np.ascontiguousarray(scores, dtype=np.float32)
    i = np.ascontiguousarray(ids, dtype=np.int64)
    out_s = np.empty((Q, k), np.float32)
    out_i = np.empty((Q, k), np.int64)
    _lib.check("vrag_topk_merge", lib.vrag_topk_merge(
        s.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p), W, Q, kk, k, 0, 0,
        out_s.ctypes.data_as(C.c_void_p), out_i.ctypes.data_as(C.c_void_p), 0, device, None))
    return out_s, out_i


class ShardComm:
    """The exchange step of sharded retrieval: one all-gather of packed `[Q, k]` lists + the merge.

    `group`: a torch.distributed process group (None = the default group; torch.distributed must be initialised).
    backend "nccl" (= RCCL on ROCm): the lists are produced, gathered and merged in HBM -- `exchange_buffers` hands the
    local search its output slots inside the packed payload, `allgather_merge_device` runs ONE `all_gather_into_tensor`
    and `vrag_topk_merge` in place on the gathered buffer, and only the merged `[Q, k]` result crosses to the host;
    backend "gloo" (CPU tests, or ranks sharing one GPU): host tensors, the merge still runs on the GPU unless `merge`
    is given (CPU-only test boxes pass `merge=merge_topk`)."""

    def __init__(self, group=None, device: int = 0, merge: Optional[Callable] = None):
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("ShardComm needs an initialised torch.distributed process group")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self.device = device
        self._merge = merge
        self._vcomm = None          # vrag_comm handle: the library's own RCCL communicator (include/vrag_amd.h)
        self._bufs = {}             # exchange buffers by (Q, k_in, k_out): allocated once, reused by every search of that shape
        if self.backend == "nccl" and merge is None:
            # every rank enters (the steps inside are collective); a rank that asks for torch's communicator (VRAG_COMM=torch) or
            # cannot load RCCL votes "no" in the preflight and ALL ranks keep torch's communicator
            self._vcomm = self._create_library_comm(dist, group)

    def _create_library_comm(self, dist, group):
        """The exchange runs behind the C ABI (`vrag_topk_allgather_merge`: ncclAllGather + merge on one stream); torch's
        process group only carries the 128-byte unique id from the group's rank 0 to the others."""
        import torch

        from . import _lib

        lib = _lib.load()
        dev = torch.device("cuda", self.device)
        # Every step below is collective, so no rank may leave it alone.  PREFLIGHT first: everything that can fail on ONE rank
        # only -- this rank wants torch's communicator (VRAG_COMM=torch), RCCL is not loadable here (vrag_comm_get_unique_id binds
        # it) -- is agreed on by all ranks (MIN) BEFORE anybody enters the blocking ncclCommInitRank; after the preflight the only
        # failures left are collective ones (ncclCommInitRank refusing), which every rank sees.
        buf = (C.c_uint8 * 128)()
        local_ok = 1
        if os.environ.get("VRAG_COMM", "") == "torch":
            local_ok = 0
            self._vcomm_error = "VRAG_COMM=torch"
        elif lib.vrag_comm_get_unique_id(buf) != 0:
            local_ok = 0
            self._vcomm_error = _lib.last_error()
        pre = torch.tensor([local_ok], dtype=torch.int32, device=dev)
        dist.all_reduce(pre, op=dist.ReduceOp.MIN, group=group)
        if int(pre.item()) != 1:
            if getattr(self, "_vcomm_error", "") != "VRAG_COMM=torch":
                import logging

                logging.getLogger(__name__).warning("library RCCL communicator unavailable (%s): the exchange uses torch.distributed",
                                                    getattr(self, "_vcomm_error", "a peer rank voted no"))
            return None
        ident = torch.zeros(129, dtype=torch.uint8, device=dev)            # [128 id bytes | 1 = valid]
        if self.rank == 0:
            ident[:128].copy_(torch.frombuffer(bytearray(buf), dtype=torch.uint8))
            ident[128] = 1
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(ident, src=src, group=group)
        host = ident.cpu().numpy()
        h = C.c_void_p()
        ok = 0
        if int(host[128]) == 1:
            ok = 1 if lib.vrag_comm_create(bytes(host[:128].tobytes()), self.rank, self.world, self.device, C.byref(h)) == 0 else 0
            if not ok:
                self._vcomm_error = _lib.last_error()
        agreed = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN, group=group)
        if int(agreed.item()) == 1:
            return h
        if ok:
            lib.vrag_comm_destroy(h)
        import logging

        logging.getLogger(__name__).warning("library RCCL communicator unavailable (%s): the exchange uses torch.distributed",
                                            getattr(self, "_vcomm_error", "a peer rank failed"))
        return None

    @property
    def exchange_backend(self) -> str:
        """Who runs the data-path collective: "vrag_comm" (RCCL behind the C ABI), "torch.distributed" or "host"."""
        if self._vcomm is not None:
            return "vrag_comm"
        return "torch.distributed" if self.on_gpu else "host"

    def close(self) -> None:
        """Destroys the library's communicator (collective-free, but call it on every rank BEFORE torch's process group is
        torn down: `GpuVectorStore.close` does).  Not left to `__del__`: at interpreter exit torch may already be gone."""
        self._bufs = {}
        if getattr(self, "_vcomm", None) is not None:
            from . import _lib

            _lib.load().vrag_comm_destroy(self._vcomm)
            self._vcomm = None

    @property
    def on_gpu(self) -> bool:
        """True when collectives take device tensors (RCCL): the exchange then never leaves HBM."""
        return self.backend == "nccl" and self._merge is None

    # ---------------------------------------------------------------- small host-side collectives of the store
    def barrier(self) -> None:
        import torch.distributed as dist

        dist.barrier(group=self.group)

    def gather_objects(self, obj) -> list:
        """Every rank's `obj`, in rank order, on every rank (result payloads of the hits a rank owns)."""
        import torch.distributed as dist

        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def union_mask(self, local: np.ndarray) -> np.ndarray:
        """Element-wise OR of every rank's boolean array (each rank marks the rows IT owns that pass a filter)."""
        import torch
        import torch.distributed as dist

        t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.uint8))
        if self.backend == "nccl":
            t = t.to(torch.device("cuda", self.device))
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t.cpu().numpy().astype(bool)

    # ---------------------------------------------------------------- the exchange
    @staticmethod
    def payload_bytes(n: int) -> int:
        return (n * 12 + 7) // 8 * 8                            # [ids i64 x n | scores f32 x n | pad]

    def exchange_buffers(self, Q: int, k: int):
        """Device payload of one exchange: (uint8 tensor, ids pointer, scores pointer).  The local search writes its
        `[Q, k]` global ids / scores straight into it (`vrag_*_index_search_device`)."""
        payload = self._exchange_set(Q, k, k)[0]
        base = payload.data_ptr()
        return payload, base, base + Q * k * 8

    def _exchange_set(self, Q: int, kk: int, k: int):
        """(payload, gathered, merged scores, merged ids) for one exchange shape -- allocated once per shape and reused: every
        use is ordered on torch's current stream and the merged lists are copied to the host before the call returns."""
        import torch

        key = (Q, kk)
        got = self._bufs.get(key)
        if got is None:
            if len(self._bufs) >= 16:
                self._bufs.clear()
            dev = torch.device("cuda", self.device)
            nbytes = self.payload_bytes(Q * kk)
            got = [torch.empty(nbytes, dtype=torch.uint8, device=dev), torch.empty(self.world * nbytes, dtype=torch.uint8, device=dev), {}]
            self._bufs[key] = got
        outs = got[2].get(k)
        if outs is None:
            dev = got[0].device
            outs = (torch.empty((Q, k), dtype=torch.float32, device=dev), torch.empty((Q, k), dtype=torch.int64, device=dev))
            got[2][k] = outs
        return got[0], got[1], outs[0], outs[1]

    def allgather_merge_device(self, payload, Q: int, kk: int, k: int) -> Tuple[np.ndarray, np.ndarray]:
        """`payload` (from `exchange_buffers`, filled on torch's current stream) of every rank -> merged `[Q, k]` on the
        host.  No host copy between the local search and the merge: gather and merge read and write HBM only."""
        import torch
        import torch.distributed as dist

        from . import _lib

        lib = _lib.load()
        n = Q * kk
        nbytes = self.payload_bytes(n)
        _, flat, d_out_s, d_out_i = self._exchange_set(Q, kk, k)
        base = flat.data_ptr()
        stream = torch.cuda.current_stream(payload.device).cuda_stream
        if self._vcomm is not None:      # THE collective of the retrieval path, behind the C ABI: ncclAllGather + merge on `stream`
            _lib.check("vrag_topk_allgather_merge", lib.vrag_topk_allgather_merge(
                self._vcomm, C.c_void_p(payload.data_ptr()), C.c_void_p(base), Q, kk, k, C.c_void_p(d_out_s.data_ptr()),
                C.c_void_p(d_out_i.data_ptr()), C.c_void_p(stream)))
            return d_out_s.cpu().numpy(), d_out_i.cpu().numpy()
        dist.all_gather_into_tensor(flat, payload, group=self.group)    # VRAG_COMM=torch: the same exchange on torch's communicator
        _lib.check("vrag_topk_merge", lib.vrag_topk_merge(
            C.c_void_p(base + n * 8), C.c_void_p(base), self.world, Q, kk, k, nbytes, nbytes,
            C.c_void_p(d_out_s.data_ptr()), C.c_void_p(d_out_i.data_ptr()), 1, self.device, C.c_void_p(stream)))
        return d_out_s.cpu().numpy(), d_out_i.cpu().numpy()

    def allgather_merge(self, scores: np.ndarray, ids: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
        """Local lists `[Q, k_in]` (GLOBAL ids, -1 = empty) held on the HOST -> merged `[Q, k]`, identical on all ranks
        (gloo groups, and lists longer than one device pass, which `vrag_*_index_search` pages through the host)."""
        import torch
        import torch.distributed as dist

        Q, kk = scores.shape
        n = Q * kk
        nbytes = self.payload_bytes(n)
        payload = np.zeros(nbytes, np.uint8)
        payload[: n * 8] = np.ascontiguousarray(ids, dtype=np.int64).view(np.uint8).reshape(-1)
        payload[n * 8: n * 12] = np.ascontiguousarray(scores, dtype=np.float32).view(np.uint8).reshape(-1)
        t = torch.from_numpy(payload)
        if self.on_gpu:
            return self.allgather_merge_device(t.to(torch.device("cuda", self.device)), Q, kk, k)
        if self.backend == "nccl":
            t = t.to(torch.device("cuda", self.device))
        flat = torch.empty(self.world * nbytes, dtype=torch.uint8, device=t.device)
        dist.all_gather_into_tensor(flat, t, group=self.group)
        gathered = flat.view(self.world, nbytes)
        if self._merge is not None:
            g = gathered.cpu().numpy()
            all_i = np.stack([g[w, : n * 8].view(np.int64).reshape(Q, kk) for w in range(self.world)])
            all_s = np.stack([g[w, n * 8: n * 12].view(np.float32).reshape(Q, kk) for w in range(self.world)])
            return self._merge(all_s, all_i, k)
        from . import _lib

        lib = _lib.load()
        out_s = np.empty((Q, k), np.float32)
        out_i = np.empty((Q, k), np.int64)
        g = np.ascontiguousarray(gathered.numpy())
        base = g.ctypes.data
        _lib.check("vrag_topk_merge", lib.vrag_topk_merge(
            C.c_void_p(base + n * 8), C.c_void_p(base), self.world, Q, kk, k, nbytes, nbytes,
            out_s.ctypes.data_as(C.c_void_p), out_i.ctypes.data_as(C.c_void_p), 0, self.device, None))
        return out_s, out_i


class ShardedTopK:
    """Wraps one shard's search over the contiguous row range starting at `shard_base`; `search` returns the global
    top-k (identical on every rank).  `local_search(queries, k) -> (scores[Q,k], local_ids[Q,k])` answers on the host;
    `shard` (a `DenseShard` / `SparseShard`, optional) lets an RCCL group keep the lists in HBM from the local search to
    the merge (`search_device` writes them into the exchange payload with `shard_base` added on the device)."""

    def __init__(self, local_search, shard_base: int, group=None, device: int = 0, merge: Optional[Callable] = None, shard=None):
        self.local_search = local_search
        self.shard_base = int(shard_base)
        self.group = group
        self.device = device
        self._merge = merge
        self._shard = shard
        self._comm: Optional[ShardComm] = None

    def search(self, queries, k: int) -> Tuple[np.ndarray, np.ndarray]:
        import torch.distributed as dist

        live = dist.is_available() and dist.is_initialized()
        if live and self._comm is None:
            self._comm = ShardComm(self.group, self.device, self._merge)
        if live and self._comm.on_gpu and self._shard is not None and k <= 64:
            import torch

            Q = len(queries)
            payload, ids_ptr, scores_ptr = self._comm.exchange_buffers(Q, k)
            stream = C.c_void_p(torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream)
            self._shard.search_device(queries, k, scores_ptr, ids_ptr, id_base=self.shard_base, stream=stream)
            return self._comm.allgather_merge_device(payload, Q, k, k)
        scores, ids = self.local_search(queries, k)
        ids = np.where(ids >= 0, ids + self.shard_base, -1).astype(np.int64)
        if not live or (self._comm.world == 1 and not self._comm.on_gpu):
            return np.where(ids >= 0, scores, -np.inf).astype(np.float32), ids
        return self._comm.allgather_merge(scores, ids, k)


def extract_spans_sharded(extractor, question: str, search_results: list, comm: "ShardComm"):
    """The extraction half of the data-parallel path: every rank holds a full extractor replica, takes its
    contiguous `shard_range` of the (question, chunk) pairs -- the units are independent, so the GPUs exchange nothing while
    they compute -- and the per-chunk span lists meet in one host-side `all_gather_object` (a few KB of strings).  Returns, on
    every rank, what `extractor.extract_spans(question, search_results)` returns on one: `{chunk text: [spans]}` in the order
    of `search_results` (the reference's contract: verbatim_core/extractors.py:233-268, call site verbatim_rag/core.py:255)."""
    lo, hi = shard_range(len(search_results), comm.rank, comm.world)
    local = extractor.extract_spans(question, search_results[lo:hi]) if hi > lo else {}
    merged: dict = {}
    for part in comm.gather_objects(local):      # rank order == chunk order; a text seen twice keeps its first entry's position
        for text, spans in part.items():
            merged.setdefault(text, spans)
    return merged
