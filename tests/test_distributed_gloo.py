"""N>1 path on CPU: world_size 2, gloo.  Sharding + ONE all-gather of per-shard top-k + merge."""
import os
import sys

import numpy as np

import verbatim_rag_amd  # noqa: F401
from verbatim_rag_amd.distributed import merge_topk, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for n, w in [(10, 3), (256, 8), (5, 8), (0, 2)]:
        parts = [shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1


def test_merge_topk_total_order():
    s = np.array([[[3.0, 2.0, -np.inf]], [[3.0, 2.5, 1.0]]], np.float32)   # [W=2, Q=1, k=3]
    i = np.array([[[7, 9, -1]], [[4, 8, 2]]], np.int64)
    ms, mi = merge_topk(s, i, 4)
    assert mi.tolist() == [[4, 7, 8, 9]] and ms.tolist() == [[3.0, 3.0, 2.5, 2.0]]   # tie 3.0 -> id asc


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import verbatim_rag_amd  # noqa: F401
    from oracle import topk_ref as T
    from verbatim_rag_amd.distributed import ShardedTopK, shard_range

    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)
    X = rng.integers(-64, 65, size=(3001, 64)).astype(np.float32) / 64     # dyadic grid: sums exact in any order
    Q = rng.integers(-64, 65, size=(5, 64)).astype(np.float32) / 64
    lo, hi = shard_range(len(X), rank, world)
    st = ShardedTopK(lambda qs, k: T.dense_topk(X[lo:hi], qs, k), shard_base=lo)
    s, i = st.search(Q, 7)
    rs, ri = T.dense_topk(X, Q, 7)
    ok = bool(np.array_equal(i, ri) and np.array_equal(s, rs))
    # sparse rows sharded the same way; query 1 matches a single document (one shard returns only -1 rows), k = 70
    # exceeds one device pass and most queries' hit counts
    V, n = 200, 1501
    lens = rng.integers(1, 6, n)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx = np.concatenate([np.sort(rng.choice(V - 1, int(m), replace=False)) for m in lens]).astype(np.int32)
    val = (rng.integers(1, 5, len(idx)) / 4).astype(np.float32)
    idx[indptr[1400]] = V - 1                                   # the only document with term V-1
    qp = np.asarray([0, 3, 4, 6], np.int64)
    qi = np.asarray([3, 50, 120, V - 1, 7, 9], np.int32)
    qv = np.asarray([1.0, 0.5, 2.0, 1.0, 0.25, 1.5], np.float32)
    lo, hi = shard_range(n, rank, world)
    sub = (indptr[lo:hi + 1] - indptr[lo], idx[indptr[lo]:indptr[hi]], val[indptr[lo]:indptr[hi]])
    sp = ShardedTopK(lambda qs, k: T.sparse_topk(*sub, V, *qs, k), shard_base=lo)
    for k in (5, 70):
        s2, i2 = sp.search((qp, qi, qv), k)
        rs2, ri2 = T.sparse_topk(indptr, idx, val, V, qp, qi, qv, k)
        ok = ok and bool(np.array_equal(i2, ri2) and np.array_equal(s2, rs2)) and i2[1, 0] == 1400 and (i2[1, 1:] == -1).all()
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_topk_world2_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]
