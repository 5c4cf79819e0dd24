"""Data-parallel sharding of the hot path across the GPUs of one node.

* Extraction / embedding: every (question, chunk) pair is independent -> `shard_range` splits the
  units contiguously over ranks, full weight replica per GPU, NO data-path collective.
* Retrieval: the corpus is row-sharded (rank r owns rows [base_r, base_r + n_r)); queries are
  replicated; each rank computes its local top-k; ONE all-gather per query batch carries
  `[Q, k]` (fp32 score, i64 id) pairs over RCCL/xGMI (a few KB..MB: latency-bound), then every
  rank merges the world_size lists with the total order (score desc, id asc).
The reference has no distributed code at all; this is new, not a translation.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def shard_range(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split, first `n % world` ranks get one extra unit."""
    q, r = divmod(n_units, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def merge_topk(scores: np.ndarray, ids: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """scores/ids: [W, Q, k] per-shard lists (id -1 = empty) -> [Q, k] merged by (score desc, id asc)."""
    W, Q, kk = scores.shape
    s = scores.transpose(1, 0, 2).reshape(Q, W * kk)
    i = ids.transpose(1, 0, 2).reshape(Q, W * kk)
    out_s = np.full((Q, k), -np.inf, np.float32)
    out_i = np.full((Q, k), -1, np.int64)
    for q in range(Q):
        valid = i[q] >= 0
        sq, iq = s[q][valid], i[q][valid]
        order = np.lexsort((iq, -sq.astype(np.float64)))[:k]
        out_s[q, : len(order)] = sq[order]
        out_i[q, : len(order)] = iq[order]
    return out_s, out_i


class ShardedTopK:
    """Wraps a local search callable `local(queries, k) -> (scores[Q,k], local_ids[Q,k])`."""

    def __init__(self, local_search, shard_base: int, group=None, device: Optional[str] = None):
        self.local_search = local_search
        self.shard_base = int(shard_base)
        self.group = group
        self.device = device

    def search(self, queries, k: int) -> Tuple[np.ndarray, np.ndarray]:
        import torch
        import torch.distributed as dist

        scores, ids = self.local_search(queries, k)
        ids = np.where(ids >= 0, ids + self.shard_base, -1).astype(np.int64)
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return merge_topk(scores[None], ids[None], k)
        world = dist.get_world_size(self.group)
        dev = self.device or ("cuda" if dist.get_backend(self.group) == "nccl" else "cpu")
        # one all-gather per query batch: pack (score bits, id) into one int64 tensor [Q, k, 2]
        payload = np.stack([scores.astype(np.float32).view(np.int32).astype(np.int64), ids], axis=-1)
        t = torch.from_numpy(payload).to(dev)
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t, group=self.group)
        g = torch.stack(gathered).cpu().numpy()
        all_scores = g[..., 0].astype(np.int32).view(np.float32)
        all_ids = g[..., 1]
        return merge_topk(all_scores, all_ids, k)
