"""N>1 path on CPU: world_size 2, gloo.  Sharding + ONE all-gather of per-shard top-k + merge, first on the bare
`ShardedTopK` wrapper, then through a row-sharded `GpuVectorStore` (the reference's `VectorStore.query` call site,
verbatim_rag/index.py:552-655 -> vector_stores/milvus_base.py:236-280).  No GPU here: the local searches are CPU
stand-ins answering from the exact oracle and the merge is the host statement `merge_topk`; tests/test_sharded_gpu.py
runs the same store test with the HIP shards and the device-side merge."""
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
    ms, mi = merge_topk(s, i, 7)                                                     # more than there is: -1 / -inf tail
    assert mi.tolist() == [[4, 7, 8, 9, 2, -1, -1]] and ms[0, 5:].tolist() == [-np.inf, -np.inf]


def test_merge_topk_matches_a_per_query_sort():
    rng = np.random.default_rng(5)
    W, Q, k = 5, 37, 9
    s = (rng.integers(-8, 9, (W, Q, k)) / 8).astype(np.float32)
    i = np.stack([rng.permutation(1000)[: Q * k].reshape(Q, k) + 1000 * w for w in range(W)]).astype(np.int64)
    i[rng.random((W, Q, k)) < 0.2] = -1
    ms, mi = merge_topk(s, i, 12)
    for q in range(Q):
        pairs = sorted(((-float(s[w, q, j]), int(i[w, q, j])) for w in range(W) for j in range(k) if i[w, q, j] >= 0))[:12]
        assert [p[1] for p in pairs] == [int(x) for x in mi[q] if x >= 0]
        assert [-p[0] for p in pairs] == [float(x) for x, y in zip(ms[q], mi[q]) if y >= 0]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    try:
        import torch.distributed as dist

        import verbatim_rag_amd  # noqa: F401
        from oracle import topk_ref as T
        from verbatim_rag_amd.distributed import ShardedTopK, merge_topk, shard_range

        dist.init_process_group("gloo", rank=rank, world_size=world)
        rng = np.random.default_rng(0)
        X = rng.integers(-64, 65, size=(3001, 64)).astype(np.float32) / 64     # dyadic grid: sums exact in any order
        Q = rng.integers(-64, 65, size=(5, 64)).astype(np.float32) / 64
        lo, hi = shard_range(len(X), rank, world)
        st = ShardedTopK(lambda qs, k: T.dense_topk(X[lo:hi], qs, k), shard_base=lo, merge=merge_topk)
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
        sp = ShardedTopK(lambda qs, k: T.sparse_topk(*sub, V, *qs, k), shard_base=lo, merge=merge_topk)
        for k in (5, 70):
            s2, i2 = sp.search((qp, qi, qv), k)
            rs2, ri2 = T.sparse_topk(indptr, idx, val, V, qp, qi, qv, k)
            ok = ok and bool(np.array_equal(i2, ri2) and np.array_equal(s2, rs2)) and i2[1, 0] == 1400 and (i2[1, 1:] == -1).all()
        q.put((rank, ok))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:   # never leave the parent waiting for its timeout
        import traceback

        q.put((rank, f"{type(exc).__name__}: {exc}\n{traceback.format_exc()}"))


def _run_world(target, world=2, timeout=240):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(60)
    return sorted(res)


def test_sharded_topk_world2_gloo():
    assert _run_world(_worker) == [(0, True), (1, True)]


def _store_worker(rank, world, port, q):
    """A row-sharded GpuVectorStore (stand-in shards, host merge) must answer exactly like a single-rank store."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    try:
        import torch.distributed as dist

        import verbatim_rag_amd  # noqa: F401
        from tests.sharded_store_cases import build_and_query, cpu_stand_ins
        from verbatim_rag_amd.distributed import ShardComm, merge_topk

        dist.init_process_group("gloo", rank=rank, world_size=world)
        with cpu_stand_ins():
            sharded = build_and_query(comm=ShardComm(merge=merge_topk))
            single = build_and_query(comm=None)
            ok = sharded == single or "sharded store answers differ from the single-rank store"
            # persistence: every rank writes its vector file, rank 0 the row table; a reload with the same world size answers alike
            import tempfile

            from verbatim_rag_amd import vector_stores as vs

            box = [tempfile.mkdtemp(prefix="vrag_shard_")] if rank == 0 else [None]
            dist.broadcast_object_list(box, src=0)
            rng = np.random.default_rng(3)
            dense = np.where(rng.random((37, 64)) < 0.5, 0.5, -0.5).astype(np.float32)
            st = vs.GpuVectorStore(dense_dim=64, enable_sparse=False, sparse_vocab=None, comm=ShardComm(merge=merge_topk))
            st.add_vectors([f"r{i}" for i in range(37)], dense.tolist(), None, [f"t{i}" for i in range(37)], [""] * 37,
                           [{"n": i} for i in range(37)])
            st.delete(["r5"])
            before = [(r.id, r.score) for r in st.query(dense_query=dense[9].tolist(), top_k=6, search_type="dense")]
            st.save(box[0])
            dist.barrier()
            st2 = vs.GpuVectorStore.load(box[0], comm=ShardComm(merge=merge_topk))
            after = [(r.id, r.score) for r in st2.query(dense_query=dense[9].tolist(), top_k=6, search_type="dense")]
            if before != after or before[0][0] != "r9" or len(st2._ids) != 36 or len(st2._owned) not in (18, 19) or len(st2._texts) != len(st2._owned):
                ok = f"save / load changed the answers: {before} vs {after}"
        q.put((rank, ok))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:
        import traceback

        q.put((rank, f"{type(exc).__name__}: {exc}\n{traceback.format_exc()}"))


def test_sharded_store_world2_gloo_equals_single_rank():
    assert _run_world(_store_worker) == [(0, True), (1, True)]


def _reshard_worker(rank, world, port, q):
    """Round 3: (a) `payload="replicated"` answers like the default sharded payload; (b) a directory saved by 2 ranks is
    opened by 2 ranks and -- in the parent -- by 1, and one saved by 1 rank is opened by 2 (re-cut on load);
    (c) a 10^7-row store: sharded ingest (5 * 10^6 rows per rank), save, load, same answers."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    try:
        import time

        import torch.distributed as dist

        import verbatim_rag_amd  # noqa: F401
        from tests.sharded_store_cases import build_and_query, cpu_stand_ins
        from verbatim_rag_amd import vector_stores as vs
        from verbatim_rag_amd.distributed import ShardComm, merge_topk

        dist.init_process_group("gloo", rank=rank, world_size=world)
        box = [os.environ["VRAG_TEST_DIR"]]
        ok = True
        with cpu_stand_ins():
            comm = ShardComm(merge=merge_topk)
            rng = np.random.default_rng(3)
            n, dim, vocab = 203, 64, 300
            dense = np.where(rng.random((n, dim)) < 0.5, 0.5, -0.5).astype(np.float32)
            sparse = [{int(t): float(v) for t, v in zip(rng.choice(vocab, 7, replace=False), rng.integers(1, 64, 7) / 64)} for _ in range(n)]

            def fill(**kw):
                st = vs.GpuVectorStore(dense_dim=dim, sparse_vocab=vocab, **kw)
                st.add_vectors([f"r{i}" for i in range(n)], dense, sparse, [f"t{i}" for i in range(n)], [f"e{i}" for i in range(n)],
                               [{"document_id": f"d{i % 5}", "n": i} for i in range(n)])
                st.delete(["r5", "r100"])
                return st

            def ask(st):
                out = []
                for kw in (dict(search_type="dense", top_k=6), dict(search_type="hybrid", top_k=5),
                           dict(search_type="sparse", top_k=4, filter='metadata["document_id"] == "d3"')):
                    res = st.query_batch(dense_queries=[dense[9].tolist(), dense[77].tolist()], sparse_queries=[sparse[9], sparse[77]], **kw)
                    out.append([[(r.id, r.score, r.text, r.enhanced_text, sorted(r.metadata.items())) for r in rs] for rs in res])
                out.append([(r.id, r.text) for r in st.query(top_k=4, filter='metadata["n"] in [7, 8, 9, 100]')])     # filter-only browse
                return out

            sharded, replicated = fill(comm=comm), fill(comm=comm, payload="replicated")
            want = ask(sharded)
            if ask(replicated) != want or len(sharded._texts) not in (101, 102) or len(replicated._texts) != n:
                ok = "replicated payload answers differ from the sharded payload"
            sharded.save(os.path.join(box[0], "by2"))                         # ends in a barrier
            if ok is True and ask(vs.GpuVectorStore.load(os.path.join(box[0], "by2"), comm=comm)) != want:
                ok = "2 ranks -> 2 ranks: answers changed"
            if rank == 0:
                single = vs.GpuVectorStore(dense_dim=dim, sparse_vocab=vocab)
                single.add_vectors([f"r{i}" for i in range(n)], dense, sparse, [f"t{i}" for i in range(n)], [f"e{i}" for i in range(n)],
                                   [{"document_id": f"d{i % 5}", "n": i} for i in range(n)])
                single.delete(["r5", "r100"])
                if ok is True and ask(single) != want:
                    ok = "single-rank store answers differ"
                single.save(os.path.join(box[0], "by1"))
                if ok is True and ask(vs.GpuVectorStore.load(os.path.join(box[0], "by2"))) != want:
                    ok = "2 ranks -> 1 rank: answers changed"
            dist.barrier()
            back = vs.GpuVectorStore.load(os.path.join(box[0], "by1"), comm=comm)
            if ok is True and (ask(back) != want or len(back._owned) not in (100, 101)):
                ok = "1 rank -> 2 ranks: answers changed"
            # (c) 10^7 rows
            big = int(os.environ.get("VRAG_TEST_BIG_ROWS", "10000000"))
            t0 = time.perf_counter()
            X = np.where(np.random.default_rng(11).random((big, 8)) < 0.5, 0.5, -0.5).astype(np.float32)
            X[12345] = 0.0
            X[12345, 0] = 1.0                                                 # the only one-hot row: its own best match
            st = vs.GpuVectorStore(dense_dim=8, enable_sparse=False, sparse_vocab=None, comm=comm)
            st.add_vectors([f"r{i}" for i in range(big)], X, None, [""] * big, [""] * big, [None] * big)
            first = [(r.id, r.score) for r in st.query(dense_query=X[12345].tolist(), top_k=3, search_type="dense")]
            st.delete([first[1][0]])
            second = [(r.id, r.score) for r in st.query(dense_query=X[12345].tolist(), top_k=3, search_type="dense")]
            st.save(os.path.join(box[0], "big"))
            st2 = vs.GpuVectorStore.load(os.path.join(box[0], "big"), comm=comm)
            third = [(r.id, r.score) for r in st2.query(dense_query=X[12345].tolist(), top_k=3, search_type="dense")]
            took = time.perf_counter() - t0
            if ok is True and not (first[0][0] == "r12345" and second[0] == first[0] and first[1] not in second and third == second
                                   and len(st2) == big - 1 and len(st2._owned) in (big // 2 - 1, big // 2)):
                ok = f"10^7-row round trip: {first} {second} {third} {len(st2)}"
            if rank == 0:
                print(f"[gloo] {big}-row ingest + 3 queries + save + load on 2 ranks: {took:.1f} s", file=sys.stderr)
        q.put((rank, ok))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:
        import traceback

        q.put((rank, f"{type(exc).__name__}: {exc}\n{traceback.format_exc()}"))


def test_sharded_store_payload_modes_resharding_and_ten_million_rows(tmp_path):
    os.environ["VRAG_TEST_DIR"] = str(tmp_path)
    try:
        assert _run_world(_reshard_worker, timeout=1500) == [(0, True), (1, True)]
    finally:
        os.environ.pop("VRAG_TEST_DIR", None)
