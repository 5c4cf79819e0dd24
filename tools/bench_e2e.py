#!/usr/bin/env python3
"""End-to-end composition probe for BASELINE configs[2] (scaled by --docs):
SPLADE-style sparse index of N synthetic chunks -> exact top-5 on the GPU -> span extraction of the
5 hits with the ModernBERT-base extractor, Q queries, cross-query batching.

Synthetic everything (no checkpoints/network): sparse rows and queries are drawn from a Zipf term
distribution (weights on the k/64 grid), chunk texts come from a pool of generated paragraphs
(tokenised once, like at ingest), encoder weights are seeded random-init.  Prints one JSON object.
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WORDS = ("the quick brown fox jumps over lazy dog tower paris iron built year tall meters visitors river city bridge "
         "stone engineer opened museum garden light night climb stairs lift wind steel design world fair paint color "
         "history france capital famous landmark ticket view north south east west floor summit restaurant glass").split()


def make_texts(rng, n, sent_per=12):
    out = []
    for _ in range(n):
        sents = []
        for _s in range(sent_per):
            ws = [WORDS[int(i)] for i in rng.integers(0, len(WORDS), size=int(rng.integers(8, 16)))]
            ws[0] = ws[0].capitalize()
            sents.append(" ".join(ws) + ".")
        out.append(" ".join(sents))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--vocab", type=int, default=30522)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--handles", type=int, default=2, help="encoder handles the extractor alternates sub-batches between")
    ap.add_argument("--hybrid", action="store_true", help="configs[4] slice: + dense 768-d bf16 shard, top-2k per method, RRF")
    ap.add_argument("--model", choices=("base", "large"), default="base", help="extractor geometry (configs[4]: large)")
    args = ap.parse_args()

    import torch
    from tokenizers import Tokenizer

    import verbatim_rag_amd  # noqa: F401
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape
    from verbatim_rag_amd.extractors import GpuModelSpanExtractor
    from verbatim_rag_amd.vector_stores import DenseShard, SparseShard, rrf_merge_rows
    from verbatim_rag_amd.weights import random_init, random_qa_head

    rng = np.random.default_rng(1234)
    n, V = args.docs, args.vocab
    nnz = np.maximum(1, rng.poisson(128, size=n))
    indptr = np.zeros(n + 1, np.int64)
    np.cumsum(nnz, out=indptr[1:])
    p = 1.0 / np.arange(1, V + 1)
    p /= p.sum()
    idx = rng.choice(V, size=int(indptr[-1]), p=p).astype(np.int32)
    val = (rng.integers(1, 193, size=int(indptr[-1])) / 64.0).astype(np.float32)
    t0 = time.perf_counter()
    shard = SparseShard(V, indptr, idx, val)
    t_build = time.perf_counter() - t0

    dense = None
    if args.hybrid:                      # bge-base-sized rows on the dyadic grid, 100k rows per upload
        t0 = time.perf_counter()
        dense = DenseShard(768, n, "bf16")
        for a in range(0, n, 100_000):
            dense.add((rng.integers(-64, 65, size=(min(100_000, n - a), 768)) / 64.0).astype(np.float32))
        t_build += time.perf_counter() - t0

    pool = make_texts(rng, 2048)
    tok = Tokenizer.from_file(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "tokenizer.json"))
    shape = ModernBertShape.base() if args.model == "base" else ModernBertShape.large()
    weights, qa_head = random_init(shape, 1234), random_qa_head(shape)
    engs = []
    for _ in range(max(1, args.handles)):
        engs.append(EncoderEngine(shape, weights, max_tokens=131072, max_seqs=2048, max_seq_len=512, max_ranges=32768,
                                  micro_batch_tokens=65536))
        engs[-1].set_qa_head(*qa_head)
    eng = engs[0]
    ext = GpuModelSpanExtractor(engine=eng, extra_engines=engs[1:], tokenizer=tok, threshold=0.5)
    ext.prepare_chunks(pool)  # ingest-time tokenisation

    Q = args.queries
    queries = [{int(t): float(v) for t, v in zip(rng.choice(V, 32, p=p), rng.integers(1, 193, 32) / 64.0)} for _ in range(Q)]
    questions = ["Where is the tall iron tower in the city?"] * Q
    dense_queries = (rng.integers(-64, 65, size=(Q, 768)) / 64.0).astype(np.float32) if args.hybrid else None

    def run():
        t = {}
        a = time.perf_counter()
        if dense is None:
            scores, ids = shard.search(queries, args.k)
        else:   # milvus_base.py:264-295: top-2k per method, equal-weight RRF (hybrid_search.py:73-129) on the host, whole batch at once
            _ss, si = shard.search(queries, 2 * args.k)
            _ds, di = dense.search(dense_queries, 2 * args.k)
            ids, _dist = rrf_merge_rows({"dense": di, "sparse": si}, args.k, {"dense": 0.5, "sparse": 0.5})
        t["search_s"] = time.perf_counter() - a
        a = time.perf_counter()
        results = [[types.SimpleNamespace(text=pool[int(i) % len(pool)]) for i in row if i >= 0] for row in ids]
        spans = ext.extract_spans_batch(questions, results)
        t["extract_s"] = time.perf_counter() - a
        return t, ids, spans

    run()  # warm-up
    # Long-lived objects (torch's module graph alone is ~200k tracked objects, a full collection of which costs
    # ~60 ms of stopped Python) go to the permanent generation, as a Python server does after start-up.
    import gc

    gc.collect()
    gc.freeze()
    torch.cuda.synchronize()
    a = time.perf_counter()
    t, ids, spans = run()
    torch.cuda.synchronize()
    total = time.perf_counter() - a
    n_pairs = int((ids >= 0).sum())
    print(json.dumps({
        "workload": f"sparse index {n} docs ({int(indptr[-1])} nnz, vocab {V})" + (f" + dense {n} x 768 bf16, RRF of top-{2 * args.k} per method" if args.hybrid else "")
                    + f" + top-{args.k} + span extraction (ModernBERT-{args.model}), {Q} queries",
        "index_build_s": t_build, "total_s": total, "queries_per_s": Q / total, "search_s": t["search_s"],
        "extract_s": t["extract_s"], "search_queries_per_s": Q / t["search_s"],
        "extract_pairs_per_s": n_pairs / t["extract_s"], "pairs": n_pairs,
        "spans_returned": int(sum(len(v) for d in spans for v in d.values())),
        "handles": len(engs),
        "note": "host-inclusive wall time (ctypes calls, packing, dict building); extraction batched across queries",
    }))
    shard.close()
    if dense is not None:
        dense.close()
    for e in engs:
        e.close()


if __name__ == "__main__":
    main()
