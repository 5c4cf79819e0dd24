"""N>1 retrieval path on the GPU: world_size 2 over gloo, both ranks on GPU 0 (the GPU box has one device;
the driver's multi-GPU run uses RCCL, same code with backend "nccl").  Local searches are the HIP shards
(`DenseShard` / `SparseShard` through the C ABI), the lists meet in ONE all-gather and are merged by
`topk_merge_shards_kernel`; the transcript must equal the single-rank GPU store's and the CPU oracle's."""
import os
import sys

import numpy as np
import pytest

import verbatim_rag_amd  # noqa: F401
from verbatim_rag_amd.distributed import merge_topk, merge_topk_device

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_device_merge_equals_host_statement():
    rng = np.random.default_rng(3)
    for W, Q, k_in, k_out in [(2, 5, 7, 7), (8, 33, 10, 10), (3, 4, 64, 100), (70, 3, 5, 12), (1, 2, 3, 3)]:
        s = (rng.integers(-16, 17, (W, Q, k_in)) / 16).astype(np.float32)
        order = np.argsort(-s, axis=2, kind="stable")
        s = np.take_along_axis(s, order, axis=2)
        ids = np.stack([np.sort(rng.choice(100000, (Q, k_in), replace=False), axis=1) + 100000 * w for w in range(W)]).astype(np.int64)
        # lists must be sorted by (score desc, id asc): ids ascending inside equal scores holds because ids ascend along the list
        tail = rng.integers(0, k_in + 1, (W, Q))
        dead = np.arange(k_in)[None, None, :] >= tail[:, :, None]
        ids[dead] = -1
        hs, hi = merge_topk(s, ids, k_out)
        ds, di = merge_topk_device(s, ids, k_out)
        assert np.array_equal(hi, di), (W, Q, k_in, k_out)
        assert np.array_equal(hs[hi >= 0], ds[di >= 0])
        assert np.all(np.isneginf(ds[di < 0]))


def test_device_merge_in_place_on_a_packed_gather_buffer():
    """on_device = 1 with list strides: the layout ShardComm hands over under RCCL ([ids | scores | pad] per rank)."""
    import ctypes as C

    import torch

    from verbatim_rag_amd import _lib

    rng = np.random.default_rng(4)
    W, Q, k = 4, 6, 5
    n = Q * k
    nbytes = (n * 12 + 7) // 8 * 8
    s = -np.sort(-(rng.integers(-16, 17, (W, Q, k)) / 16).astype(np.float32), axis=2)
    ids = np.stack([np.sort(rng.choice(5000, (Q, k), replace=False), axis=1) + 5000 * w for w in range(W)]).astype(np.int64)
    buf = np.zeros((W, nbytes), np.uint8)
    for w in range(W):
        buf[w, : n * 8] = ids[w].view(np.uint8).reshape(-1)
        buf[w, n * 8: n * 12] = s[w].view(np.uint8).reshape(-1)
    d = torch.from_numpy(buf).cuda()
    out_s = torch.empty((Q, k), dtype=torch.float32, device="cuda")
    out_i = torch.empty((Q, k), dtype=torch.int64, device="cuda")
    _lib.check("vrag_topk_merge", _lib.load().vrag_topk_merge(
        C.c_void_p(d.data_ptr() + n * 8), C.c_void_p(d.data_ptr()), W, Q, k, k, nbytes, nbytes, C.c_void_p(out_s.data_ptr()),
        C.c_void_p(out_i.data_ptr()), 1, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    hs, hi = merge_topk(s, ids, k)
    assert np.array_equal(out_i.cpu().numpy(), hi) and np.array_equal(out_s.cpu().numpy(), hs)


def _store_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    try:
        import torch.distributed as dist

        import verbatim_rag_amd  # noqa: F401
        from tests.sharded_store_cases import build_and_query
        from verbatim_rag_amd.distributed import ShardComm

        dist.init_process_group("gloo", rank=rank, world_size=world)
        out = {}
        for dtype in ("f32", "bf16"):
            sharded = build_and_query(comm=ShardComm(device=0), dense_dtype=dtype)     # HIP shards + device merge
            single = build_and_query(comm=None, dense_dtype=dtype)                    # one GPU holds everything
            out[dtype] = (sharded == single, sharded if rank == 0 else None)
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:
        import traceback

        q.put((rank, f"{type(exc).__name__}: {exc}\n{traceback.format_exc()}"))


def test_sharded_gpu_store_world2_equals_single_rank_and_oracle():
    import torch.multiprocessing as mp

    from tests.sharded_store_cases import build_and_query, cpu_stand_ins

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_store_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
    for rank, out in res:
        assert isinstance(out, dict), out
        assert out["f32"][0] and out["bf16"][0], f"rank {rank}: sharded != single-rank"
    with cpu_stand_ins():
        oracle = build_and_query(comm=None)            # exact CPU top-k behind the same host logic
    # unit rows of this data set are dyadic with 2 significant bits: exact in bf16 too, so both dtypes equal the oracle
    assert res[0][1]["f32"][1] == oracle
    assert res[0][1]["bf16"][1] == oracle


def _extract_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    try:
        import json
        import types

        import torch.distributed as dist
        from tokenizers import Tokenizer

        import verbatim_rag_amd  # noqa: F401
        from oracle import modernbert_np as O
        from verbatim_rag_amd.distributed import ShardComm, extract_spans_sharded
        from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape
        from verbatim_rag_amd.extractors import GpuModelSpanExtractor

        dist.init_process_group("gloo", rank=rank, world_size=world)
        G = os.path.join(ROOT, "tests", "golden")
        tiny = dict(vocab_size=512, hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=192,
                    pad_token_id=0, cls_token_id=1, sep_token_id=2)
        w = O.random_weights(O.EncoderConfig(**tiny), seed=7)
        z = np.load(os.path.join(G, "encoder_tiny.npz"))
        eng = EncoderEngine(ModernBertShape(**tiny), w, max_tokens=8192, max_seqs=64, max_seq_len=512, max_ranges=1024)
        eng.set_qa_head(z["qa_Wc"], z["qa_bc"])
        ext = GpuModelSpanExtractor(engine=eng, tokenizer=Tokenizer.from_file(os.path.join(G, "tokenizer.json")), threshold=0.45)
        with open(os.path.join(G, "host_fixtures.json")) as f:
            fx = json.load(f)
        texts = []
        for run in fx["extract_e2e"]["runs"]:
            texts += [t for t in run["texts"] if t not in texts]
        results = [types.SimpleNamespace(text=t) for t in texts]
        question = fx["extract_e2e"]["runs"][0]["question"]
        out = {}
        # A rank's share has fewer rows than the whole call and may take another launch-bound GEMM configuration (K split over
        # four waves / one-wave tiles): since round 6 these sum in ONE order (csrc/gemm_bf16.hip, KCH), so the logits -- and with
        # them every threshold decision -- are bit-identical whatever the share (round 5 had to pin one configuration here).
        for n in (len(results), 5, 1):                       # 1 pair over 2 ranks: rank 1's shard is empty
            sharded = extract_spans_sharded(ext, question, results[:n], ShardComm(device=0))
            single = ext.extract_spans(question, results[:n])
            out[n] = (sharded == single and list(sharded) == list(single), len(single), sum(len(v) for v in single.values()))
        eng.close()
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:
        import traceback

        q.put((rank, f"{type(exc).__name__}: {exc}\n{traceback.format_exc()}"))


def test_extraction_split_world2_equals_the_single_rank_call():
    """The OTHER half of the N > 1 path: `shard_range` over the (question, chunk) pairs, one extractor
    replica per rank (both on GPU 0 here), span dicts gathered -- equal to one rank extracting everything, keys in chunk order."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_extract_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
    for rank, out in res:
        assert isinstance(out, dict), out
        assert all(v[0] for v in out.values()), (rank, out)
        assert out[max(out)][1] >= 4, out                 # the fixture really holds several chunks
