"""The RCCL branch of the retrieval exchange on the one GPU the test box has (north_star: "RCCL all-gather over xGMI of
per-shard top-k for the final merge"): a world-1 process group with the `nccl` backend drives exactly the code the
8-GPU run takes -- `vrag_*_index_search_device` writes the local `[Q, k]` lists (global rows through the device row
table) into the packed payload in HBM, ONE `all_gather_into_tensor`, `vrag_topk_merge` in place on the gathered buffer,
and only the merged lists are copied to the host.  Results must equal the host-list exchange over a gloo group created
in the same process, the single-GPU store and the CPU oracle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    try:
        import torch
        import torch.distributed as dist

        import verbatim_rag_amd  # noqa: F401
        from oracle import topk_ref as T
        from tests.sharded_store_cases import build_and_query, cpu_stand_ins
        from verbatim_rag_amd import vector_stores as vs
        from verbatim_rag_amd.distributed import ShardComm, ShardedTopK

        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        out = {}
        comm = ShardComm(device=0)
        out["backend"] = (comm.backend, comm.on_gpu)

        # 1. the exchange primitives: device-resident lists -> all-gather -> in-place merge == the host search
        rng = np.random.default_rng(0)
        X = rng.standard_normal((50_000, 384)).astype(np.float32)
        Q = rng.standard_normal((33, 384)).astype(np.float32)
        sh = vs.DenseShard(384, len(X), "f32", 0)
        sh.add(X)
        hs, hi = sh.search(Q, 10)
        table = torch.arange(1000, 1000 + len(X), dtype=torch.int64, device="cuda")          # local row -> global row
        payload, ids_ptr, scores_ptr = comm.exchange_buffers(len(Q), 10)
        stream = torch.cuda.current_stream().cuda_stream
        sh.search_device(Q, 10, scores_ptr, ids_ptr, row_map=table.data_ptr(), n_map=len(X), stream=stream)
        ms, mi = comm.allgather_merge_device(payload, len(Q), 10, 10)
        out["exchange_backend"] = comm.exchange_backend       # RCCL behind the C ABI (vrag_comm_create / vrag_topk_allgather_merge)
        # the same payload through torch's communicator (VRAG_COMM=torch) gives the same lists
        os.environ["VRAG_COMM"] = "torch"
        comm_t = ShardComm(device=0)
        os.environ.pop("VRAG_COMM")
        mst, mit = comm_t.allgather_merge_device(payload, len(Q), 10, 10)
        out["torch_comm_equals_library_comm"] = bool(comm_t.exchange_backend == "torch.distributed" and np.array_equal(ms, mst)
                                                     and np.array_equal(mi, mit))
        # raw all-gather entry point + communicator info
        from verbatim_rag_amd import _lib as L
        import ctypes as CT

        rk, wd, ver = CT.c_int32(-1), CT.c_int32(-1), CT.c_int32(0)
        L.check("info", L.load().vrag_comm_info(comm._vcomm, CT.byref(rk), CT.byref(wd), CT.byref(ver)))
        src = torch.arange(1000, dtype=torch.uint8, device="cuda")
        dst = torch.zeros(1000, dtype=torch.uint8, device="cuda")
        L.check("ag", L.load().vrag_comm_allgather(comm._vcomm, CT.c_void_p(src.data_ptr()), CT.c_void_p(dst.data_ptr()), 1000,
                                                    CT.c_void_p(stream)))
        torch.cuda.synchronize()
        out["raw_allgather"] = bool((rk.value, wd.value) == (0, 1) and ver.value > 0 and torch.equal(src, dst))
        rs, ri = T.dense_topk(X, Q, 10)
        out["dense_device_exchange"] = bool(np.array_equal(mi, hi + 1000) and np.array_equal(ms, hs) and np.array_equal(hi, ri)
                                            and np.array_equal(hs, rs))
        # a table shorter than the shard: rows beyond it are reported as missing (-1 / -inf), never as garbage
        payload2, ids2, scores2 = comm.exchange_buffers(2, 10)
        sh.search_device(Q[:2], 10, scores2, ids2, row_map=table.data_ptr(), n_map=10, stream=stream)
        torch.cuda.synchronize()
        raw = payload2.cpu().numpy()
        mi2, ms2 = raw[:160].view(np.int64), raw[160:240].view(np.float32)
        out["short_table"] = bool(((mi2 == -1) | ((mi2 >= 1000) & (mi2 < 1010))).all() and np.isneginf(ms2[mi2 == -1]).all())
        sh.close()
        n_docs, vocab = 30_000, 2000
        lens = rng.integers(1, 30, n_docs)
        ip = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ix = np.concatenate([np.sort(rng.choice(vocab, int(m), replace=False)) for m in lens]).astype(np.int32)
        vv = (rng.integers(1, 193, len(ix)) / 64.0).astype(np.float32)
        sp = vs.SparseShard(vocab, ip, ix, vv, 0)
        queries = [{int(t): float(w) for t, w in zip(rng.choice(vocab, 6, replace=False), rng.integers(1, 64, 6) / 64)} for _ in range(19)]
        hs, hi = sp.search(queries, 7)
        payload, ids_ptr, scores_ptr = comm.exchange_buffers(19, 7)
        sp.search_device(queries, 7, scores_ptr, ids_ptr, id_base=5000, stream=stream)
        ms, mi = comm.allgather_merge_device(payload, 19, 7, 7)
        rs, ri = T.sparse_topk(ip, ix, vv, vocab, *vs.dicts_to_csr(queries), 7)
        out["sparse_device_exchange"] = bool(np.array_equal(mi, np.where(hi >= 0, hi + 5000, -1)) and np.array_equal(ms, hs)
                                             and np.array_equal(hi, ri) and np.array_equal(hs, rs))
        sp.close()
        # an empty contribution (a rank that holds no rows)
        payload, ids_ptr, scores_ptr = comm.exchange_buffers(4, 5)
        from verbatim_rag_amd import _lib
        import ctypes as C

        _lib.check("fill", _lib.load().vrag_topk_fill_empty(C.c_void_p(scores_ptr), C.c_void_p(ids_ptr), 20, 0, C.c_void_p(stream)))
        ms, mi = comm.allgather_merge_device(payload, 4, 5, 5)
        out["empty_contribution"] = bool((mi == -1).all() and np.isneginf(ms).all())

        # 2. ShardedTopK (bench.py's sharded leg) under RCCL: host lists in, device merge
        X = (rng.integers(-64, 65, size=(20_000, 128)) / 64.0).astype(np.float32)
        Q = (rng.integers(-64, 65, size=(9, 128)) / 64.0).astype(np.float32)
        sh = vs.DenseShard(128, len(X), "f32", 0)
        sh.add(X)
        rs, ri = T.dense_topk(X, Q, 6)
        s, i = ShardedTopK(lambda qs, k: sh.search(qs, k), shard_base=0, device=0).search(Q, 6)        # host lists, device merge
        out["sharded_topk_host_lists"] = bool(np.array_equal(i, ri) and np.array_equal(s, rs))
        s, i = ShardedTopK(sh.search, shard_base=700, device=0, shard=sh).search(Q, 6)                 # lists never leave HBM
        out["sharded_topk_device_lists"] = bool(np.array_equal(i, ri + 700) and np.array_equal(s, rs))
        sh.close()

        # 3. the public store, distributed=True, through the on_gpu branch: inserts in several batches, deletes, filters
        # (masked subset shards with their own device tables), k = 70 (paged through the host), a late row (sparse tail
        # segment: two device lists merged before the exchange)
        transcripts = {}
        for dtype in ("f32", "bf16"):
            transcripts[dtype] = build_and_query(comm=ShardComm(device=0), dense_dtype=dtype)
            out[f"store_{dtype}_nccl_equals_single"] = transcripts[dtype] == build_and_query(comm=None, dense_dtype=dtype)
        gloo_group = dist.new_group(backend="gloo")
        out["store_nccl_equals_gloo"] = build_and_query(comm=ShardComm(group=gloo_group, device=0)) == transcripts["f32"]
        with cpu_stand_ins():
            oracle = build_and_query(comm=None)
        out["store_equals_oracle"] = oracle == transcripts["f32"] and oracle == transcripts["bf16"]   # dyadic data: exact in both
        st = vs.GpuVectorStore(dense_dim=64, sparse_vocab=300, distributed=True)
        out["store_distributed_flag"] = (st._comm is not None and st._comm.on_gpu, st._world)
        q.put(out)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:
        import traceback

        q.put(f"{type(exc).__name__}: {exc}\n{traceback.format_exc()}")


def test_rccl_world1_exchange_and_store():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(33500 + (os.getpid() % 2000), q))
    p.start()
    out = q.get(timeout=900)
    p.join(120)
    assert isinstance(out, dict), out
    assert out.pop("backend") == ("nccl", True)
    assert out.pop("exchange_backend") == "vrag_comm"
    assert out.pop("store_distributed_flag") == (True, 1)
    bad = {k: v for k, v in out.items() if v is not True}
    assert not bad, bad
