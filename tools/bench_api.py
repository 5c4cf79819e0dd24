#!/usr/bin/env python3
"""Public-API composition probe: HotPathIndex (ingest + query_batch) over a GpuVectorStore, the GPU span extractor and
StaticVerbatimPipeline.query_batch, i.e. what a caller of the reference's VerbatimIndex / VerbatimRAG touches, with
host-inclusive wall time per stage.  The embedding providers are synthetic (unit dense rows, Zipf sparse rows: a
random-init SPLADE model would emit dense rows) -- their device path is tools/bench_embed.py's subject.
Prints one JSON object."""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


class SyntheticDense:
    def __init__(self, dim, seed):
        self.dim, self.rng = dim, np.random.default_rng(seed)

    def _rows(self, n):
        x = self.rng.standard_normal((n, self.dim)).astype(np.float32)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).tolist()

    def embed_text(self, text):
        return self._rows(1)[0]

    def embed_batch(self, texts):
        return self._rows(len(texts))

    embed_queries = embed_batch

    def get_dimension(self):
        return self.dim


class SyntheticSparse:
    def __init__(self, vocab, nnz, seed):
        self.vocab, self.nnz, self.rng = vocab, nnz, np.random.default_rng(seed)
        p = 1.0 / np.arange(1, vocab + 1)
        self.p = p / p.sum()

    def _rows(self, n, nnz):
        t = self.rng.choice(self.vocab, size=(n, nnz), p=self.p)
        v = self.rng.integers(1, 193, size=(n, nnz)) / 64.0
        return [dict(zip(ti.tolist(), vi.tolist())) for ti, vi in zip(t, v)]

    def embed_text(self, text):
        return self._rows(1, 32)[0]

    def embed_batch(self, texts):
        return self._rows(len(texts), self.nnz)

    def embed_queries(self, texts):
        return self._rows(len(texts), 32)

    def get_dimension(self):
        return self.vocab


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=20000)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--k", type=int, default=5)
    args = ap.parse_args()

    from tokenizers import Tokenizer

    import bench_e2e
    import verbatim_rag_amd  # noqa: F401
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape
    from verbatim_rag_amd.extractors import GpuModelSpanExtractor
    from verbatim_rag_amd.index import HotPathIndex
    from verbatim_rag_amd.pipeline import StaticVerbatimPipeline
    from verbatim_rag_amd.vector_stores import GpuVectorStore
    from verbatim_rag_amd.weights import random_init, random_qa_head

    rng = np.random.default_rng(7)
    pool = bench_e2e.make_texts(rng, 2048)
    n, V = args.docs, 30522
    store = GpuVectorStore(dense_dim=args.dim, sparse_vocab=V)
    index = HotPathIndex(store, dense_provider=SyntheticDense(args.dim, 1), sparse_provider=SyntheticSparse(V, 64, 2))
    t0 = time.perf_counter()
    index.add_chunks([f"c{i}" for i in range(n)], [pool[i % len(pool)] for i in range(n)],
                     metadatas=[{"document_id": f"d{i // 50}", "title": f"T{i // 50}", "source": "synthetic"} for i in range(n)])
    t_ingest = time.perf_counter() - t0

    shape = ModernBertShape.base()
    weights, head = random_init(shape, 1234), random_qa_head(shape)
    engs = []
    for _ in range(2):
        engs.append(EncoderEngine(shape, weights, max_tokens=131072, max_seqs=2048, max_seq_len=512, max_ranges=32768,
                                  micro_batch_tokens=65536))
        engs[-1].set_qa_head(*head)
    tok = Tokenizer.from_file(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "tokenizer.json"))
    ext = GpuModelSpanExtractor(engine=engs[0], extra_engines=engs[1:], tokenizer=tok, threshold=0.5)
    ext.prepare_chunks(pool)
    pipe = StaticVerbatimPipeline(index, ext, k=args.k)

    questions = [f"Where is the tall iron tower number {i} in the city?" for i in range(args.queries)]
    t0 = time.perf_counter()
    index.query_batch(questions[:8], k=args.k)                   # first query after the inserts: flush to HBM
    t_flush = time.perf_counter() - t0
    pipe.query_batch(questions[:64])                             # warm-up
    gc.collect()
    gc.freeze()
    out = {}
    for label, kw in (("hybrid", {}), ("hybrid_filtered", {"filter": 'metadata["document_id"] == "d7"'})):
        t0 = time.perf_counter()
        hits = index.query_batch(questions, k=args.k, **kw)
        t_search = time.perf_counter() - t0
        t0 = time.perf_counter()
        resp = pipe.query_batch(questions, **kw)
        t_total = time.perf_counter() - t0
        out[label] = {"retrieval_s": t_search, "retrieval_queries_per_s": len(questions) / t_search, "pipeline_s": t_total,
                      "pipeline_queries_per_s": len(questions) / t_total, "hits_per_query": float(np.mean([len(h) for h in hits])),
                      "citations_per_query": float(np.mean([len(r.structured_answer.citations) for r in resp]))}
    t0 = time.perf_counter()
    for q in questions[:100]:
        pipe.query(q)
    t_single = (time.perf_counter() - t0) / 100
    print(json.dumps({"workload": f"GpuVectorStore {n} chunks (dense {args.dim} {store.dense_dtype} + sparse vocab {V}), hybrid top-{args.k}, "
                                  f"ModernBERT-base extraction, {args.queries} queries via StaticVerbatimPipeline.query_batch",
                      "ingest_s": t_ingest, "flush_and_first_query_s": t_flush, **out, "single_query_ms": t_single * 1e3}))
    for e in engs:
        e.close()


if __name__ == "__main__":
    main()
