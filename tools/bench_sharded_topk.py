#!/usr/bin/env python3
"""BASELINE configs[3] shape: dense brute-force top-k over a corpus row-sharded across the ranks of one node, per-shard
top-k on the GPU, ONE all-gather of [Q, k] (score, id) pairs, merge on every rank (`ShardedTopK`).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_sharded_topk.py
    (VRAG_BENCH_BACKEND=gloo lets several ranks share one GPU: harness self-test on a 1-GPU box)

Rank r owns rows [r*n, (r+1)*n) of a synthetic dyadic-grid corpus; queries are identical on every rank.  Rank 0
checks the merged ids against a single-shard search over the same rows when they fit (--verify) and prints one JSON."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rows_of(rank, n, dim):
    rng = np.random.default_rng(1000 + rank)
    return (rng.integers(-64, 65, size=(n, dim)) / 64.0).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows-per-rank", type=int, default=1_250_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--verify", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("VRAG_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import verbatim_rag_amd  # noqa: F401
    from verbatim_rag_amd.distributed import ShardedTopK
    from verbatim_rag_amd.vector_stores import DenseShard

    n, dim = args.rows_per_rank, args.dim
    shard = DenseShard(dim, n, "bf16", device=local_rank)
    chunk = 125_000
    X = rows_of(rank, n, dim)
    for a in range(0, n, chunk):
        shard.add(X[a:a + chunk])
    Q = (np.random.default_rng(7).integers(-64, 65, size=(args.queries, dim)) / 64.0).astype(np.float32)
    topk = ShardedTopK(lambda q, k: shard.search(q, k), shard_base=rank * n,
                       device="cuda" if backend == "nccl" else "cpu")
    s, i = topk.search(Q, args.k)            # warm-up
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        s, i = topk.search(Q, args.k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    if world > 1:
        tt = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ok = None
    if args.verify and rank == 0:
        # every shard's rows are reproducible from its rank: search them one by one on this GPU and merge on the host
        from verbatim_rag_amd.distributed import merge_topk

        parts_s, parts_i = [], []
        for r in range(world):
            sh = DenseShard(dim, n, "bf16", device=local_rank)
            Xr = rows_of(r, n, dim)
            for a in range(0, n, chunk):
                sh.add(Xr[a:a + chunk])
            ps, pi = sh.search(Q, args.k)
            sh.close()
            parts_s.append(ps)
            parts_i.append(np.where(pi >= 0, pi + r * n, -1))
        rs, ri = merge_topk(np.stack(parts_s), np.stack(parts_i).astype(np.int64), args.k)
        ok = bool(np.array_equal(ri, i) and np.array_equal(rs, s))
    if rank == 0:
        print(json.dumps({"workload": f"dense top-{args.k}, {world} shards x {n} x {dim} bf16 rows, {args.queries} queries per batch",
                          "n_gpus": world, "backend": backend, "ms_per_batch": dt * 1e3,
                          "queries_per_s": args.queries / dt, "rows_scanned_per_s": world * n * args.queries / dt,
                          "merged_equals_per_shard_searches": ok}))
    shard.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
