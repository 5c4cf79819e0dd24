#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on the hot path: query x chunk span-extractions/sec @512 tok.

A "step" = one pass of the extractor hot path (ModernBERT-base encoder forward + per-sentence
classification head) over one batch of 256 synthetic (question, chunk) sequences of exactly
512 tokens (BASELINE.json configs[1]), inputs already resident in HBM.  N>1: one process per
GPU (torch.distributed / RCCL only for the barrier + max-over-ranks), every rank processes its
own 256 chunks (weak scaling, no data-path collective: the units are independent).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEQ = 512
N_SENT = 16
Q_TOK = 24
PROFILE_EVERY = 5          # timed region: HIP events around every 5th launch of each kernel class (every launch costs the step 1.1 %); 5 is coprime with the 44 / 16 / 28 launches a class has per step, so the sampled layers rotate from step to step
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0     # HBM3E spec peak (same guide; ~6.3 TB/s is what a streaming copy reaches)


def _gemm_source_sha16() -> str:
    """Identity of the GEMM kernels a PMC traffic summary belongs to (tools/pmc_traffic.py stamps the same hash)."""
    import hashlib

    h = hashlib.sha256()
    for f in ("gemm_bf16.hip", "gemm_bf16.h", "common.h", "qkv_attn.hip", "qkv_attn.h"):
        with open(os.path.join(ROOT, "verbatim-rag_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def synth_batch(shape, n_seqs: int, seed: int):
    """[CLS] q(24) ([SEP] sentence)x16 [SEP]  == 512 tokens; sentence lengths 30x6 + 29x10."""
    rng = np.random.default_rng(seed)
    lens = [30] * 6 + [29] * 10
    seqs, bounds = [], []
    for _ in range(n_seqs):
        ids = [shape.cls_token_id] + rng.integers(1000, 50000, size=Q_TOK).tolist()
        b = []
        for ln in lens:
            ids.append(shape.sep_token_id)
            start = len(ids)
            ids.extend(rng.integers(1000, 50000, size=ln).tolist())
            b.append((start, len(ids) - 1))
        ids.append(shape.sep_token_id)
        assert len(ids) == SEQ
        seqs.append(np.asarray(ids, dtype=np.int32))
        bounds.append(b)
    return seqs, bounds


def gemm_flops_per_step(shape, tokens: int):
    H, I = shape.hidden_size, shape.intermediate_size
    L = shape.num_hidden_layers
    return {
        "gemm_qkv": 2.0 * tokens * 3 * H * H * L,
        "gemm_wo": 2.0 * tokens * H * H * L,
        "gemm_wi": 2.0 * tokens * 2 * I * H * L,
        "gemm_wo_mlp": 2.0 * tokens * H * I * L,
    }


def chunk_flops(shape, S: int = SEQ) -> float:
    """SURVEY 8(d) algorithmic FLOP per 512-token chunk (1.222e11 for base)."""
    H, I, L = shape.hidden_size, shape.intermediate_size, shape.num_hidden_layers
    lin = L * 2.0 * S * (H * 3 * H + H * H + H * 2 * I + I * H)
    n_glob = len([l for l in range(L) if l % shape.global_attn_every_n_layers == 0])
    glob = n_glob * 4.0 * S * S * H
    loc = (L - n_glob) * 4.0 * S * (2 * (shape.local_attention // 2) + 1) * H
    return lin + glob + loc


def _cpu_threads() -> int:
    try:
        from threadpoolctl import threadpool_info

        return int(max([p.get("num_threads", 1) for p in threadpool_info()] + [1]))
    except Exception:
        return os.cpu_count() or 1


def _oracle_cfg(shape):
    from oracle import modernbert_np as O

    return O.EncoderConfig(
        vocab_size=shape.vocab_size, hidden_size=shape.hidden_size, num_hidden_layers=shape.num_hidden_layers,
        num_attention_heads=shape.num_attention_heads, intermediate_size=shape.intermediate_size,
        global_attn_every_n_layers=shape.global_attn_every_n_layers, local_attention=shape.local_attention,
        global_rope_theta=shape.global_rope_theta, local_rope_theta=shape.local_rope_theta, norm_eps=shape.norm_eps,
        pad_token_id=shape.pad_token_id, cls_token_id=shape.cls_token_id, sep_token_id=shape.sep_token_id)


def _hf_cpu_model(shape, weights):
    """`transformers` ModernBertModel (fp32, CPU) carrying the bench weights -- the module the reference's
    QAModel.forward calls (extractor_models/model.py:51,75).  Parameter init is skipped (everything is overwritten
    by load_state_dict); returns None when transformers / torch cannot provide it."""
    try:
        import torch
        import torch.nn as nn
        from transformers import ModernBertConfig, ModernBertModel
    except Exception:
        return None
    cfg = ModernBertConfig(
        vocab_size=shape.vocab_size, hidden_size=shape.hidden_size, num_hidden_layers=shape.num_hidden_layers,
        num_attention_heads=shape.num_attention_heads, intermediate_size=shape.intermediate_size,
        global_attn_every_n_layers=shape.global_attn_every_n_layers, local_attention=shape.local_attention,
        norm_eps=shape.norm_eps, pad_token_id=shape.pad_token_id, attn_implementation="sdpa")
    rp = getattr(cfg, "rope_parameters", None) or {}
    if (rp.get("full_attention", {}).get("rope_theta", shape.global_rope_theta) != shape.global_rope_theta or
            rp.get("sliding_attention", {}).get("rope_theta", shape.local_rope_theta) != shape.local_rope_theta):
        return None
    saved = (nn.Linear.reset_parameters, nn.Embedding.reset_parameters, ModernBertModel._init_weights)
    try:
        nn.Linear.reset_parameters = lambda self: None
        nn.Embedding.reset_parameters = lambda self: None
        ModernBertModel._init_weights = lambda self, module: None
        model = ModernBertModel(cfg).eval()
    finally:
        nn.Linear.reset_parameters, nn.Embedding.reset_parameters, ModernBertModel._init_weights = saved
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()},
                                                strict=False)
    if missing or unexpected:
        return None
    return model


def cpu_baseline(shape, weights, qa_w, qa_b, seqs, bounds, budget_s: float):
    """CPU leg on a bounded sample of the same batch, B = 1 per chunk like the reference's loop (extractors.py:233-268).
    Preferred: the reference's own arithmetic -- `transformers` ModernBertModel fp32 on the host cores -- plus the
    restated sentence head (oracle/); fallback: the numpy oracle end to end.  Both are checkers, never the product."""
    from oracle import modernbert_np as O

    model = _hf_cpu_model(shape, weights)
    logits, n = [], 0
    if model is not None:
        import torch

        # B = 1 x 512 tokens does not scale past ~16 threads (MI355X host, tools/cpu_threads_probe.py: 4.9 / 7.9 / 5.0 /
        # 0.9 chunks/s at 8 / 16 / 32 / 128 threads): use the best setting rather than torch's all-cores default
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        with torch.no_grad():
            model(input_ids=torch.from_numpy(np.asarray(seqs[0], dtype=np.int64))[None])   # first-call warm-up, untimed
            t0 = time.perf_counter()
            for s, b in zip(seqs, bounds):
                hid = model(input_ids=torch.from_numpy(np.asarray(s, dtype=np.int64))[None]).last_hidden_state[0].numpy()
                logits.append(O.qa_sentence_logits(hid, b, qa_w, qa_b))
                n += 1
                if time.perf_counter() - t0 > budget_s:
                    break
        dt = time.perf_counter() - t0
        multi_threads = int(torch.get_num_threads())
        # SURVEY 8(d) asks for two host figures: the best multi-thread setting above and ONE thread (what a scalar port of
        # the loop would be compared with).  Same module, same chunks, a third of the budget, at least two chunks.
        single = None
        try:
            torch.set_num_threads(1)
            n1, t1 = 0, time.perf_counter()
            with torch.no_grad():
                for s in seqs[:max(2, n)]:
                    model(input_ids=torch.from_numpy(np.asarray(s, dtype=np.int64))[None])
                    n1 += 1
                    if n1 >= 2 and time.perf_counter() - t1 > budget_s / 3:
                        break
            single = {"value": n1 / (time.perf_counter() - t1), "unit": "chunks/s", "cores": 1,
                      "sample": f"the first {n1} of the same chunks, torch.set_num_threads(1)"}
        except Exception as exc:   # a reported baseline, never a reason to lose the line
            single = {"error": repr(exc)}
        finally:
            torch.set_num_threads(multi_threads)
        return {
            "value": n / dt, "unit": "chunks/s", "cores": multi_threads, "kind": "port",
            "sample": f"{n} of the 256 synthetic 512-token chunks; transformers ModernBertModel fp32 on CPU (the module the "
                      "reference's QAModel.forward runs) + restated sentence head, B=1 per chunk like the reference loop",
            "single_thread": single,
        }, logits
    cfg = _oracle_cfg(shape)
    t0 = time.perf_counter()
    for s, b in zip(seqs, bounds):
        hid = O.encoder_forward(cfg, weights, s)
        logits.append(O.qa_sentence_logits(hid, b, qa_w, qa_b))
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {
        "value": n / dt, "unit": "chunks/s", "cores": _cpu_threads(), "kind": "port",
        "sample": f"{n} of the 256 synthetic 512-token chunks, numpy fp32 oracle (B=1 per chunk like the reference loop)",
    }, logits


def topk_recall_check(seed: int = 1234):
    """Second half of BASELINE.json's metric ("top-k recall vs CPU ref"): a small dense and a small sparse shard on the
    GPU against the exact CPU top-k (oracle/topk_ref.c -- the checker, like the cpu_baseline leg).  Never raises."""
    try:
        from oracle import topk_ref as T
        from verbatim_rag_amd.vector_stores import DenseShard, SparseShard

        rng = np.random.default_rng(seed)
        n, dim, vocab, nq, k = 20000, 128, 3000, 16, 10
        X = (rng.integers(-64, 65, size=(n, dim)) / 64.0).astype(np.float32)
        Q = (rng.integers(-64, 65, size=(nq, dim)) / 64.0).astype(np.float32)
        sh = DenseShard(dim, n, "bf16")
        sh.add(X)
        _s, di = sh.search(Q, k)
        sh.close()
        _rs, dri = T.dense_topk(X, Q, k)
        lens = rng.integers(1, 40, n)
        indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        idx = np.concatenate([np.sort(rng.choice(vocab, int(m), replace=False)) for m in lens]).astype(np.int32)
        val = (rng.integers(1, 193, len(idx)) / 64.0).astype(np.float32)
        qp = np.arange(0, 8 * (nq + 1), 8).astype(np.int64)
        qi = np.concatenate([np.sort(rng.choice(vocab, 8, replace=False)) for _ in range(nq)]).astype(np.int32)
        qv = (rng.integers(1, 193, len(qi)) / 64.0).astype(np.float32)
        sp = SparseShard(vocab, indptr, idx, val)
        _s2, si = sp.search_csr(qp, qi, qv, k)
        sp.close()
        _rs2, sri = T.sparse_topk(indptr, idx, val, vocab, qp, qi, qv, k)

        def recall(got, ref):
            return float(np.mean([len(set(g[g >= 0]) & set(r[r >= 0])) / max(1, int((r >= 0).sum())) for g, r in zip(got, ref)]))

        return {"k": k, "dense_recall": recall(di, dri), "sparse_recall": recall(si, sri),
                "indices_equal": bool(np.array_equal(di, dri) and np.array_equal(si, sri)),
                "sample": f"{nq} queries over {n} x {dim} bf16 rows and {n} sparse docs (vocab {vocab}), exact CPU top-k as reference"}
    except Exception as exc:  # the bench line must not depend on this check
        return {"error": f"{type(exc).__name__}: {exc}"}


def power_limited_rate(shape, rows: int, device: int, value: float):
    """What the matrix pipe sustains on THIS part under its package power cap: the bare main loop of the widest GEMM
    (EPI_NONE: no epilogue, nothing stored) on random operands, looped alone.  Every kernel class of the step runs at the cap
    (profiles/r03_energy_by_class.json), so this -- not the nominal 2.5 PFLOP/s -- is the rate the main loops are bound by;
    reported beside the nominal fraction, never instead of it.  Never raises."""
    try:
        import ctypes as C

        from verbatim_rag_amd import _lib as L

        ms = C.c_float()
        L.check_debug("gemm", L.load_debug().vrag_debug_gemm_ms(7, rows, 3 * shape.hidden_size, shape.hidden_size, 60, device, C.byref(ms)))
        plateau = 2.0 * rows * 3 * shape.hidden_size * shape.hidden_size / (ms.value * 1e-3) / 1e12
        return {"power_limited_mainloop_tflops": plateau, "model_frac_of_power_limited_rate": value * chunk_flops(shape) / 1e12 / plateau}
    except Exception as exc:
        return {"power_limited_mainloop_tflops": f"{type(exc).__name__}: {exc}"}


def api_leg(eng, shape, n_chunks: int, steps: int):
    """What a caller of the plug point gets (reference call site verbatim_rag/core.py:255):
    `GpuModelSpanExtractor.extract_spans_batch` for ONE question over `n_chunks` real-text chunks of ~512 tokens -- tokenise
    the question, assemble the packed ids from the ingest-time chunk cache, H2D, encoder + sentence head, read-back, softmax
    and threshold select -- against the device-resident rate of the headline metric.  Never raises."""
    try:
        import types

        from tokenizers import Tokenizer

        from verbatim_rag_amd.extractors import GpuModelSpanExtractor

        tok = Tokenizer.from_file(os.path.join(ROOT, "tests", "golden", "tokenizer.json"))
        words = [w for w, _i in sorted(tok.get_vocab().items(), key=lambda kv: kv[1]) if w.isalpha() and len(w) > 2]
        rng = np.random.default_rng(99)
        question = "where is the old tower that the engineer built near the river bridge"
        chunks = []
        for i in range(n_chunks):              # 16 sentences of 27 words + a number: ~30 tokens each, ~500 with the question
            sents = [" ".join(rng.choice(words, 27).tolist()) + f" {i}." for _ in range(N_SENT)]
            chunks.append(" ".join(s.capitalize() for s in sents))
        ext = GpuModelSpanExtractor(engine=eng, tokenizer=tok, threshold=0.5)
        results = [types.SimpleNamespace(text=c) for c in chunks]
        # the synthetic chunks run a few tokens over the 512-token budget: the packer drops their last sentence and says so
        # per chunk and call (the reference's own warning text) -- thousands of lines that pushed the JSON line out of a
        # `tail`; here the warning is counted instead and reported in the leg
        import logging

        class _Count(logging.Filter):
            n = 0

            def filter(self, record):
                if "token budget" in record.getMessage():
                    _Count.n += 1
                    return False
                return True

        xlog = logging.getLogger("verbatim_rag_amd.extractors")
        filt = _Count()
        xlog.addFilter(filt)
        all_sents, samples = ext.pack_qa(question, chunks[:8])
        tokens = float(np.mean([len(smp.input_ids) for smp in samples if smp is not None]))
        out = ext.extract_spans_batch([question], [results])      # warm-up: fills the chunk cache (ingest-time work)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = ext.extract_spans_batch([question], [results])
        dt = (time.perf_counter() - t0) / steps
        xlog.removeFilter(filt)
        return {"api_chunks_per_s": n_chunks / dt, "ms_per_call": dt * 1e3, "chunks_per_call": n_chunks, "mean_tokens_per_pair": tokens,
                "budget_warnings_suppressed": _Count.n,
                "spans_returned": int(sum(len(v) for v in out[0].values())),
                "what": "extract_spans_batch(1 question, 256 results): question tokenisation, ids from the chunk cache, H2D, "
                        "encoder + sentence head, read-back, threshold select"}
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}"}


def e2e_text_leg(shape, weights, qa_w, qa_b, device: int, n_docs: int = 1_000_000, n_queries: int = 1000, k: int = 5):
    """BASELINE configs[2] composed with TEXT queries (reference path verbatim_rag/index.py:592-655 ->
    vector_stores/milvus_base.py:250-259 -> core.py:237-277): `n_queries` question texts -> GpuSpladeProvider.embed_queries (the
    QUERY-side encoder forward on the GPU: BERT-base shape, 12 layers, V = 30 522, fp16 operands + split-operand SPLADE head,
    the provider's defaults) -> exact sparse top-k over a 10^6-document SELL index -> extract_spans_batch over the top-5 chunks of
    every question (ModernBERT-base sentence classifier, cross-query batching).  Synthetic everything: Zipf corpus generated on
    the GPU, seeded random-init encoders; the random-init head's active terms are mapped onto the corpus' Zipf distribution on the
    host (a trained SPLADE model's query terms follow it by themselves) and its decoder bias is calibrated to ~32 active terms
    per question.  Parity of the same composition: tests/test_e2e_text_in_gpu.py.  Host-inclusive wall time.  Never raises."""
    try:
        import gc
        import types

        import torch
        from tokenizers import Tokenizer

        from verbatim_rag_amd.embedding_providers import GpuSpladeProvider
        from verbatim_rag_amd.engine import BertEncoderEngine, BertShape, EncoderEngine
        from verbatim_rag_amd.extractors import GpuModelSpanExtractor
        from verbatim_rag_amd.vector_stores import SparseShard
        from verbatim_rag_amd.weights import random_init_bert

        V = 30522
        dev = torch.device("cuda", device)
        g = torch.Generator(device=dev)
        g.manual_seed(1234)
        t0 = time.perf_counter()
        nnz = torch.poisson(torch.full((n_docs,), 128.0, device=dev), generator=g).clamp_(min=1).to(torch.int64)
        indptr = torch.zeros(n_docs + 1, dtype=torch.int64, device=dev)
        torch.cumsum(nnz, 0, out=indptr[1:])
        total = int(indptr[-1])
        p = 1.0 / torch.arange(1, V + 1, device=dev, dtype=torch.float64)
        cdf = torch.cumsum(p / p.sum(), 0).to(torch.float32)
        idx = torch.searchsorted(cdf, torch.rand(total, device=dev, generator=g)).clamp_(max=V - 1).to(torch.int32)
        val = (torch.randint(1, 193, (total,), device=dev, generator=g).to(torch.float32) / 64.0)
        shard = SparseShard(V, indptr.cpu().numpy(), idx.cpu().numpy(), val.cpu().numpy())
        zipf_map = torch.searchsorted(cdf, torch.rand(V, device=dev, generator=g)).clamp_(max=V - 1).cpu().numpy()   # term id -> Zipf-distributed id
        del nnz, indptr, idx, val
        t_index = time.perf_counter() - t0

        tok = Tokenizer.from_file(os.path.join(ROOT, "tests", "golden", "tokenizer.json"))
        words = [w for w, _i in sorted(tok.get_vocab().items(), key=lambda kv: kv[1]) if w.isalpha() and len(w) > 2]
        rng = np.random.default_rng(2024)
        pool = []
        for _ in range(2048):                      # chunk texts: 12 sentences of 8-15 words, tokenised once (ingest-time work)
            sents = [" ".join(rng.choice(words, int(rng.integers(8, 16))).tolist()).capitalize() + "." for _s in range(12)]
            pool.append(" ".join(sents))
        questions = ["Where is the " + " ".join(rng.choice(words, int(rng.integers(4, 9))).tolist()) + "?" for _ in range(n_queries)]

        bshape = BertShape.bert_base()
        W = random_init_bert(bshape, seed=1234)
        emb = BertEncoderEngine(bshape, {k_: v for k_, v in W.items() if not k_.startswith("mlm.")}, max_tokens=32768, max_seqs=2048,
                                max_seq_len=128, max_ranges=2048, device=device, operand_dtype="f16")
        head = (W["mlm.dense.w"], W["mlm.dense.b"], W["mlm.ln.w"], W["mlm.ln.b"])
        emb.set_mlm_head_ex(*head, np.zeros(V, np.float32), W.get("mlm.dec.w"))
        prov = GpuSpladeProvider(emb, tok, max_length=128, sparse_cap=4096)
        rows = prov._rows(questions[:64])          # calibration: decoder bias for ~32 active terms per question
        raw = np.where(rows > 0, np.expm1(rows), 0.0)          # relu(logit) back from log1p
        thr = float(np.quantile(raw, 1.0 - 32.0 / V))
        emb.set_mlm_head_ex(*head, np.full(V, -thr, np.float32), W.get("mlm.dec.w"))

        ext_eng = EncoderEngine(shape, weights, max_tokens=131072, max_seqs=2048, max_seq_len=SEQ, max_ranges=32768,
                                micro_batch_tokens=65536, device=device)
        ext_eng.set_qa_head(qa_w, qa_b)
        ext = GpuModelSpanExtractor(engine=ext_eng, tokenizer=tok, threshold=0.5)
        ext.prepare_chunks(pool)

        def run():
            t = {}
            a = time.perf_counter()
            dicts = prov.embed_queries(questions)                                   # query-side encoder + SPLADE head on the GPU
            queries = []
            for d in dicts:                                                         # synthetic-data plumbing: ids onto the corpus' Zipf law
                q = {}
                for term, w_ in d.items():
                    z = int(zipf_map[term])
                    q[z] = max(q.get(z, 0.0), w_)
                queries.append(q or {0: 1.0})
            t["embed_s"] = time.perf_counter() - a
            a = time.perf_counter()
            _scores, ids = shard.search(queries, k)
            t["search_s"] = time.perf_counter() - a
            a = time.perf_counter()
            results = [[types.SimpleNamespace(text=pool[int(i) % len(pool)]) for i in row if i >= 0] for row in ids]
            spans = ext.extract_spans_batch(questions, results)
            t["extract_s"] = time.perf_counter() - a
            return t, ids, spans, queries

        run()
        gc.collect()
        torch.cuda.synchronize()
        a = time.perf_counter()
        t, ids, spans, queries = run()
        torch.cuda.synchronize()
        total_s = time.perf_counter() - a
        n_pairs = int((ids >= 0).sum())
        # the same chain for ONE question at a time (the reference's VerbatimRAG.query: core.py:237-277), host-inclusive
        def run_one(qtext):
            a0 = time.perf_counter()
            d = prov.embed_queries([qtext])[0]
            q1 = {}
            for term, w_ in d.items():
                z = int(zipf_map[term])
                q1[z] = max(q1.get(z, 0.0), w_)
            a1 = time.perf_counter()
            _s1, ids1 = shard.search([q1 or {0: 1.0}], k)
            a2 = time.perf_counter()
            res1 = [types.SimpleNamespace(text=pool[int(i) % len(pool)]) for i in ids1[0] if i >= 0]
            ext.extract_spans(qtext, res1)
            a3 = time.perf_counter()
            return a1 - a0, a2 - a1, a3 - a2
        for qtext in questions[:8]:
            run_one(qtext)
        singles = np.asarray([run_one(qtext) for qtext in questions[8:72]])
        single = {"ms_per_query_median": float(np.median(singles.sum(axis=1)) * 1e3), "embed_ms": float(np.median(singles[:, 0]) * 1e3),
                  "search_ms": float(np.median(singles[:, 1]) * 1e3), "extract_ms": float(np.median(singles[:, 2]) * 1e3), "queries": len(singles)}
        out = {"e2e_queries_per_s": n_queries / total_s, "single_query": single, "total_s": total_s, "embed_s": t["embed_s"], "search_s": t["search_s"],
               "extract_s": t["extract_s"], "pairs": n_pairs, "index_build_s": t_index,
               "mean_query_terms": float(np.mean([len(q) for q in queries])),
               "spans_returned": int(sum(len(v) for d in spans for v in d.values())),
               "what": f"{n_queries} question texts -> SPLADE query encoder on the GPU (BERT-base shape, V = {V}, fp16 operands, split head) -> "
                       f"exact sparse top-{k} over {n_docs} docs ({total} nnz) -> extract_spans_batch over the top-{k} chunks "
                       "(ModernBERT-base sentence classifier); host-inclusive wall time, one process"}
        shard.close()
        emb.close()
        ext_eng.close()
        return out
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}"}


def token_head_f16_leg(shape, weights, seqs, micro_batch_tokens: int, device: int, steps: int):
    """The reference's DEFAULT extractor (v2 highlighter, extractors.py:203-228) runs the token-classification head; here
    with fp16 MFMA operands (what keeps per-token logits within 1e-3) + the split-operand head, same 256 x 512 batch,
    resident inputs.  Informational: the headline metric is the sentence classifier in bf16.  Never raises."""
    try:
        import torch

        from verbatim_rag_amd.engine import EncoderEngine

        n = len(seqs)
        eng = EncoderEngine(shape, weights, max_tokens=n * SEQ, max_seqs=n, max_seq_len=SEQ, max_ranges=64,
                            micro_batch_tokens=micro_batch_tokens, device=device, operand_dtype="f16")
        rng = np.random.default_rng(7)
        H = shape.hidden_size
        eng.set_token_head((rng.standard_normal((H, H)) * 0.02).astype(np.float32), np.ones(H, np.float32),
                           (rng.standard_normal((2, H)) * H ** -0.5).astype(np.float32), np.zeros(2, np.float32))
        stream = torch.cuda.current_stream().cuda_stream
        eng.load_batch(seqs, stream)
        for _ in range(2):
            eng.run(stream)
            eng.run_token_head(stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.run(stream)
            eng.run_token_head(stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        sat = eng.f16_saturated(reset=True)
        eng.close()
        return {"chunks_per_s": n / dt, "ms_per_step": dt * 1e3, "operand_dtype": "f16", "f16_saturated": bool(sat),
                "what": "encoder + token-classification head (v2 highlighter arithmetic), 256 x 512 tokens, resident inputs"}
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}"}


def ragged_lengths(rows_per_micro_batch: int, n_micro_batches: int, seed: int = 4242, lo: int = 64, hi: int = SEQ, mean: float = 200.0,
                   aligned: bool = False) -> np.ndarray:
    """Seeded pair lengths in [lo, hi] with mean ~`mean` (what the reference's 512-CHARACTER chunker produces once the question
    is prepended: verbatim_rag/chunker_providers.py:531-572 cuts ~100-150-token windows; longer ones come from the markdown
    chunkers) -- a shifted, clipped gamma.
    aligned=False (the reported figure): ANY length, drawn until the batch's packed rows (tokens + the 8-row alignment gap a
    pair of unaligned length leaves) would pass the budget of the headline batch -- the engine's equal-size micro-batch cuts and
    tile-round remainders are whatever they come out as, like a real chunker's output.
    aligned=True (best case, reported beside it): lengths are multiples of 8 (no alignment gaps) and the last pair of every
    micro-batch takes exactly the rows that are left, so every micro-batch is a whole number of GEMM tile rounds."""
    rng = np.random.default_rng(seed)
    out = []
    if not aligned:
        budget, rows = rows_per_micro_batch * n_micro_batches, 0
        while True:
            n = int(np.clip(lo + rng.gamma(2.0, (mean - lo) / 2.0), lo, hi))
            packed = (n + 7) // 8 * 8
            if rows + packed > budget:
                break
            out.append(n)
            rows += packed
        return np.asarray(out, np.int32)
    for _ in range(n_micro_batches):
        rows = 0
        while True:
            left = rows_per_micro_batch - rows
            if left <= hi:
                out.append(left)
                break
            n = int(np.clip(lo + rng.gamma(2.0, (mean - lo) / 2.0), lo, hi)) // 8 * 8   # multiples of 8: no alignment gaps, so the
            if left - n < lo:                          # batch fills its rows in any order; no remainder shorter than the shortest pair
                n = max(lo, n - lo - 8)
            out.append(n)
            rows += n
    return np.asarray(out, np.int32)


def synth_ragged_batch(shape, lens, seed: int):
    """[CLS] q(24) ([SEP] sentence)* [SEP] of the given total lengths: sentences of ~30 tokens, the last one shorter."""
    rng = np.random.default_rng(seed)
    seqs, bounds = [], []
    for n in lens.tolist():
        ids = [shape.cls_token_id] + rng.integers(1000, 50000, size=Q_TOK).tolist()
        b = []
        while len(ids) < n - 3:
            ln = min(30, n - 2 - len(ids))
            ids.append(shape.sep_token_id)
            start = len(ids)
            ids.extend(rng.integers(1000, 50000, size=ln).tolist())
            b.append((start, len(ids) - 1))
        ids.extend([shape.sep_token_id] * (n - len(ids)))
        seqs.append(np.asarray(ids, dtype=np.int32))
        bounds.append(b)
    return seqs, bounds


def ragged_leg(shape, weights, qa_w, qa_b, tokens: int, micro_batch_tokens: int, device: int, steps: int, headline_tokens_per_s: float,
               aligned: bool = False):
    """The headline step on the pair lengths real chunkers produce: the same packed rows per step (131 072,
    two full micro-batches) as pairs of 64-512 tokens, mean ~200, inputs resident, encoder + sentence head.  Reported beside the
    headline as chunks/s and as tokens/s (real tokens, alignment gaps not counted) relative to the 512-token batch.  Never raises."""
    try:
        import torch

        from verbatim_rag_amd.engine import EncoderEngine

        mbt = micro_batch_tokens or tokens
        lens = ragged_lengths(mbt, max(1, tokens // mbt), aligned=aligned)
        seqs, bounds = synth_ragged_batch(shape, lens, seed=77)
        n_rng = sum(len(b) for b in bounds)
        eng = EncoderEngine(shape, weights, max_tokens=int(lens.sum()) + 8 * len(lens), max_seqs=len(lens), max_seq_len=SEQ,
                            max_ranges=n_rng, micro_batch_tokens=micro_batch_tokens, device=device)
        eng.set_qa_head(qa_w, qa_b)
        stream = torch.cuda.current_stream().cuda_stream

        eng.load_batch(seqs, stream)
        eng.load_ranges(np.repeat(np.arange(len(seqs), dtype=np.int32), [len(b) for b in bounds]),
                        np.asarray([r[0] for b in bounds for r in b], np.int32), np.asarray([r[1] for b in bounds for r in b], np.int32), stream)
        for _ in range(2):
            eng.run(stream)
            eng.run_qa_head(stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.run(stream)
            eng.run_qa_head(stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        eng.set_concurrency(1)                  # untimed: the kernels of this batch one at a time, per class
        eng.run(stream)
        torch.cuda.synchronize()
        eng.read_profile(reset=True)
        eng.set_profiling(1)
        eng.run(stream)
        eng.run_qa_head(stream)
        torch.cuda.synchronize()
        eng.set_profiling(0)
        iso = {k: {"avg_launch_us": v[0] / v[1] * 1e3, "ms_per_step": v[0], "launches": v[1]} for k, v in eng.read_profile(reset=True).items() if v[1] > 0}
        eng.close()
        tps = float(lens.sum()) / dt
        return {"ragged_chunks_per_s": len(lens) / dt, "ms_per_step": dt * 1e3, "chunks_per_step": int(len(lens)), "tokens_per_step": int(lens.sum()),
                "length_min_mean_max": [int(lens.min()), float(lens.mean()), int(lens.max())], "tokens_per_s": tps,
                "tokens_per_s_vs_512_token_batch": tps / headline_tokens_per_s, "single_stream_pass_by_class": iso,
                "order": "as drawn (the engine packs the sequences of a micro-batch into the fused kernel's 8-slot groups itself, best fit decreasing)",
                "lengths": "multiples of 8, every micro-batch filled to a whole number of tile rounds (best case)" if aligned else
                           "unconstrained (alignment gaps and tile-round remainders as they fall)",
                "what": "encoder + sentence head over pairs of 64-512 tokens (seeded, mean ~200) within the packed-row budget of the headline batch (8-row alignment gaps between pairs are rows, not tokens), resident inputs"}
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}"}


def _grid_rows(n: int, dim: int, seed: int) -> np.ndarray:
    """`[n, dim]` fp32 rows on the dyadic grid k / 64 (dot products exact in fp32 in any order), filled slab by slab."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, dim), np.float32)
    for a in range(0, n, 65536):
        b = min(n, a + 65536)
        np.multiply(rng.integers(-64, 65, size=(b - a, dim), dtype=np.int8), np.float32(1.0 / 64.0), out=out[a:b])
    return out


def sharded_retrieval_leg(rank: int, world: int, device: int, backend: str, rows_per_rank: int = 1_250_000, dim: int = 768,
                          nq: int = 1024, k: int = 10, reps: int = 5):
    """Second TIMED leg of an N > 1 run: the exchange step of the path at BASELINE configs[3]'s per-GPU size (10^7 x 768 bf16
    rows over 8 GPUs = 1.25 * 10^6 per rank; north_star: "RCCL all-gather of per-shard top-k for the final merge").  Every
    rank holds its own row shard (generated ON its GPU: dyadic-grid rows from a seeded device generator, handed to the index
    with vrag_dense_index_add_device -- no host RNG, no upload), answers the replicated batch of `nq` queries on it (tiled
    batched search, lists left in HBM by vrag_dense_index_search_device), ONE all-gather carries the `[nq, k]` lists
    (vrag_topk_allgather_merge: ncclAllGather + merge behind the C ABI on RCCL groups), every GPU merges.  Timed like the
    headline: barrier + synchronize on both sides, `reps` exchanges, MAX over ranks.  Returns a report on every rank (rank 0
    prints it); a failure is reported on stderr AND in the line, never as a crash."""
    import torch
    import torch.distributed as dist

    report = {"rows_per_rank": None}
    try:
        from verbatim_rag_amd.distributed import ShardedTopK, merge_topk
        from verbatim_rag_amd.vector_stores import DenseShard

        rows_per_rank = int(os.environ.get("VRAG_BENCH_SHARD_ROWS", rows_per_rank))
        dev = torch.device("cuda", device)
        shard = DenseShard(dim, rows_per_rank, "bf16", device)
        gen = torch.Generator(device=dev)
        gen.manual_seed(7700 + rank)
        t_gen = time.perf_counter()
        for a in range(0, rows_per_rank, 131072):       # 131 072-row slabs: 400 MB of fp32 staging at a time
            b = min(rows_per_rank, a + 131072)
            slab = torch.randint(-64, 65, (b - a, dim), generator=gen, device=dev, dtype=torch.int32).to(torch.float32) / 64.0
            shard.add_device(slab.data_ptr(), b - a, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            del slab
        t_gen = time.perf_counter() - t_gen
        Q = _grid_rows(nq, dim, 5)                                                               # replicated queries: same on every rank
        topk = ShardedTopK(shard.search, shard_base=rank * rows_per_rank, device=device, shard=shard)
        s, i = topk.search(Q, k)                                                                 # warm-up + the checked result
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            topk.search(Q, k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        dist.barrier()
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item()) / reps
        ls, li = shard.search(Q, k)                                                              # the host-side statement of the same lists
        gathered = [None] * world
        dist.all_gather_object(gathered, (ls, np.where(li >= 0, li + rank * rows_per_rank, -1)))
        comm = topk._comm
        shard.close()
        hs, hi = merge_topk(np.stack([g[0] for g in gathered]), np.stack([g[1] for g in gathered]).astype(np.int64), k)
        report = {"sharded_queries_per_s": nq / dt, "ms_per_batch": dt * 1e3, "rows_total": rows_per_rank * world,
                  "rows_per_rank": rows_per_rank, "dim": dim, "rows_dtype": "bf16", "queries": nq, "k": k, "reps": reps,
                  "shard_generated_on_gpu_s": t_gen,
                  "collective": f"one all-gather of {nq * k * 12} B per rank, merge on every GPU",
                  "exchange_backend": comm.exchange_backend if comm is not None else None,
                  "process_group_backend": dist.get_backend(), "world_size_reported_by_backend": int(dist.get_world_size()),
                  "lists_device_resident": bool(comm is not None and comm.on_gpu),
                  "merged_equals_host_merge_of_shard_lists": bool(np.array_equal(hi, i) and np.array_equal(hs, s))}
    except Exception as exc:
        import traceback

        print(f"bench.py: sharded retrieval leg FAILED on rank {rank}: {type(exc).__name__}: {exc}\n{traceback.format_exc()}",
              file=sys.stderr, flush=True)
        report = {"error": f"rank {rank}: {type(exc).__name__}: {exc}"}
    try:   # one rank's failure must show in rank 0's line
        every = [None] * world
        dist.all_gather_object(every, report.get("error"))
        errs = [e for e in every if e]
        if errs:
            report["error"] = "; ".join(errs)
    except Exception:
        pass
    return report


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chunks", type=int, default=256, help="sequences per step per GPU")
    ap.add_argument("--micro-batch-tokens", type=int, default=int(os.environ.get("VRAG_MICRO_BATCH", "65536")))
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU-baseline work (0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--operand-dtype", choices=["bf16", "f16"], default="bf16",
                    help="MFMA operand type (bf16 = the metric's dtype; f16 = the highlighter extractor's default, informational)")
    ap.add_argument("--model", choices=["base", "large"], default="base",
                    help="base = BASELINE configs[1] (the metric); large = ModernBERT-large geometry (configs[4] extractor), informational")
    args = ap.parse_args()
    if os.environ.get("VRAG_BENCH_SKIP_LEGS"):   # A/B sessions (tools/ab_streams.sh): the headline and its roofline only
        args.cpu_budget = 0.0

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher the driver would have used (one rank per GPU)
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]])

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # VRAG_BENCH_BACKEND=gloo: harness self-test on a box with fewer GPUs than ranks (ranks share devices, the
    # barrier / max-reduce go through gloo on CPU tensors); the real multi-GPU run uses RCCL ("nccl").
    backend = os.environ.get("VRAG_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import verbatim_rag_amd  # noqa: F401
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape
    from verbatim_rag_amd.weights import random_init, random_qa_head

    shape = ModernBertShape.base() if args.model == "base" else ModernBertShape.large()
    weights = random_init(shape, seed=1234)
    qa_w, qa_b = random_qa_head(shape)
    n_chunks = args.chunks
    eng = EncoderEngine(shape, weights, max_tokens=n_chunks * SEQ, max_seqs=n_chunks, max_seq_len=SEQ,
                        max_ranges=n_chunks * N_SENT, micro_batch_tokens=args.micro_batch_tokens, device=local_rank,
                        operand_dtype=args.operand_dtype)
    eng.set_qa_head(qa_w, qa_b)
    seqs, bounds = synth_batch(shape, n_chunks, seed=1234 + rank)
    stream = torch.cuda.current_stream().cuda_stream

    # inputs resident in HBM before the timed region
    eng.load_batch(seqs, stream)
    seq_idx = np.repeat(np.arange(n_chunks, dtype=np.int32), N_SENT)
    st = np.asarray([b[0] for bs in bounds for b in bs], dtype=np.int32)
    en = np.asarray([b[1] for bs in bounds for b in bs], dtype=np.int32)
    eng.load_ranges(seq_idx, st, en, stream)
    torch.cuda.synchronize()

    def step():
        eng.run(stream)
        eng.run_qa_head(stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    eng.read_profile(reset=True)
    eng.set_profiling(0 if args.no_profile else PROFILE_EVERY)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    per_rank_ms, backend_world = [elapsed / args.steps * 1e3], 1
    if world > 1:
        dist.barrier()
        dev = "cuda" if backend == "nccl" else "cpu"
        mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                     # every rank's own clock, so a straggler is visible in the one JSON line
        per_rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        backend_world = int(dist.get_world_size())
        tt = mine.clone()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    eng.set_profiling(False)
    prof = eng.read_profile(reset=True)
    logits = eng.read_qa_logits(stream)

    # Second, untimed pass with the micro-batches serialised on ONE stream: per-kernel durations
    # without the cross-stream overlap of the timed region (kernel quality, not job throughput).
    iso = None
    if not args.no_profile and rank == 0:
        eng.set_concurrency(1)
        eng.run(stream)
        torch.cuda.synchronize()
        eng.set_profiling(True)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        eng.set_profiling(False)
        iso = eng.read_profile(reset=True)
        eng.set_concurrency(2)

    sharded = None   # N > 1: runs AFTER rank 0 has assembled the line (below), under a watchdog

    if rank == 0:
        total_chunks = world * n_chunks * args.steps
        value = total_chunks / elapsed
        fl = gemm_flops_per_step(shape, n_chunks * SEQ)
        roof = None
        breakdown = {k: {"ms_per_step": v[0] / max(1, args.steps), "launches_per_step": v[1] / max(1, args.steps)}
                     for k, v in prof.items() if v[1] > 0}
        # Kernel classes by rocprofv3 kernel name: both residual GEMMs (attention Wo, mlp Wo) are ONE instantiation.
        # the fused kernel's class carries the Wqkv product AND the attention of its layers
        n_glob = len([l for l in range(shape.num_hidden_layers) if l % shape.global_attn_every_n_layers == 0])
        tok = n_chunks * SEQ
        fl["qkv_attn_global"] = fl["gemm_qkv"] * n_glob / shape.num_hidden_layers + n_glob * 4.0 * tok * SEQ * shape.hidden_size
        fl["qkv_attn_local"] = fl["gemm_qkv"] * (1 - n_glob / shape.num_hidden_layers) + \
            (shape.num_hidden_layers - n_glob) * 4.0 * tok * (2 * (shape.local_attention // 2) + 1) * shape.hidden_size
        classes = {
            "vrag::qkv_attn_kernel (Wqkv + RoPE + attention per sequence and head)": ["qkv_attn_global", "qkv_attn_local"],
            "vrag::gemm_bf16_kernel<EPI_QKV_ROPE> (Wqkv + RoPE + V^T)": ["gemm_qkv"],
            "vrag::gemm_bf16_kernel<EPI_RESIDUAL> (attn Wo + mlp Wo, fp32 residual RMW)": ["gemm_wo", "gemm_wo_mlp"],
            "vrag::gemm_bf16_kernel<EPI_GEGLU> (Wi + GeGLU)": ["gemm_wi"],
        }

        def class_stats(pr, steps):
            out = {}
            for name, keys in classes.items():
                ms = sum(pr.get(k, (0.0, 0))[0] for k in keys)
                n = sum(pr.get(k, (0.0, 0))[1] for k in keys)
                fl_tot = sum(fl[k] for k in keys) * steps
                if n > 0 and ms > 0:
                    out[name] = {"tflops": fl_tot / (ms * 1e-3) / 1e12, "avg_launch_ms": ms / n, "ms_per_step": ms / steps,
                                 "flop_per_launch": fl_tot / n}
            return out

        timed_cls = class_stats(prof, args.steps)
        if timed_cls:
            dom = max(timed_cls, key=lambda c: timed_cls[c]["ms_per_step"])       # largest summed time in the TIMED region
            t = timed_cls[dom]
            traffic, traffic_src = None, None
            cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")) \
                if os.path.isdir(os.path.join(ROOT, "profiles")) else []
            # PMC passes cannot run inside this process: a committed summary of tools/profile_round.sh.  A summary measured on OTHER
            # kernels is not this build's traffic: it must carry the hash of the kernel sources it was taken from
            # (tools/pmc_traffic.py) and that hash must be the current one.
            for cand in reversed(cands):
                try:
                    tj = json.load(open(os.path.join(ROOT, "profiles", cand)))
                except Exception:
                    continue
                vals = [tj[k]["hbm_bytes_per_launch"] for k in classes[dom] if k in tj and tj[k].get("hbm_bytes_per_launch")]
                if vals and tj.get("_gemm_source_sha16") == _gemm_source_sha16():
                    traffic, traffic_src = float(np.mean(vals)), cand
                    break
                if vals and traffic_src is None:
                    traffic_src = f"{cand} is stale (kernel sources changed since): re-run tools/profile_round.sh"
            # algorithmic HBM bytes per launch of the class: the residual GEMMs are the sum of an MFMA
            # main loop and an HBM-speed fp32 read-modify-write epilogue, so both fractions are reported
            M_mb = min(args.micro_batch_tokens or n_chunks * SEQ, n_chunks * SEQ)
            Hs, Is = shape.hidden_size, shape.intermediate_size
            alg_bytes = {"qkv_attn_global": M_mb * Hs * 2 + 3 * Hs * Hs * 2 + M_mb * Hs * 2,
                         "qkv_attn_local": M_mb * Hs * 2 + 3 * Hs * Hs * 2 + M_mb * Hs * 2,
                         "gemm_qkv": M_mb * Hs * 2 + 3 * Hs * Hs * 2 + 3 * M_mb * Hs * 2,
                         "gemm_wi": M_mb * Hs * 2 + 2 * Is * Hs * 2 + M_mb * Is * 2,
                         # residual GEMMs, split stream: A operand + weights + the operand plane and the byte remainder plane in and
                         # out (round 6: 6 B per element; round 4's fp16 remainder plane: 8)
                         "gemm_wo": M_mb * Hs * 2 + Hs * Hs * 2 + 2 * M_mb * Hs * 3,
                         "gemm_wo_mlp": M_mb * Is * 2 + Hs * Is * 2 + 2 * M_mb * Hs * 3}
            dom_bytes = float(np.mean([alg_bytes[k] for k in classes[dom]]))
            gbps = dom_bytes / (t["avg_launch_ms"] * 1e-3) / 1e9
            frac_mfma, frac_hbm = t["tflops"] / PEAK_BF16_TFLOPS, gbps / PEAK_HBM_GBPS
            roof = {
                "bound": "hbm" if frac_hbm > frac_mfma else "mfma", "kernel": dom,
                "achieved": gbps if frac_hbm > frac_mfma else t["tflops"], "peak": PEAK_HBM_GBPS if frac_hbm > frac_mfma else PEAK_BF16_TFLOPS,
                "unit": "GB/s" if frac_hbm > frac_mfma else "TFLOP/s", "frac": max(frac_mfma, frac_hbm),
                "frac_mfma": frac_mfma, "frac_hbm": frac_hbm, "achieved_tflops": t["tflops"], "achieved_gbps": gbps,
                "algorithmic_bytes_per_launch": dom_bytes, "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": t["avg_launch_ms"], "flop_per_launch": t["flop_per_launch"],
                "phase": f"timed region: HIP events recorded on the launch stream around every {PROFILE_EVERY}th launch of the class "
                         "(vrag_encoder_set_profiling), averaged over the timed steps",
                "all_classes_timed_region": {c: {"tflops": v["tflops"], "frac": v["tflops"] / PEAK_BF16_TFLOPS,
                                                 "avg_launch_ms": v["avg_launch_ms"]} for c, v in timed_cls.items()},
            }
            # the step-level fraction is the headline fraction (one driver-timed number, comparable round to round); `frac` above
            # is the dominant class under two-stream co-residency, `isolated_frac` the same class alone on the chip
            roof["model_mfma_frac"] = value * chunk_flops(shape) / 1e12 / (PEAK_BF16_TFLOPS * world)
            if iso:
                iso_cls = class_stats(iso, 2)
                if dom in iso_cls:
                    roof["isolated_frac"] = iso_cls[dom]["tflops"] / PEAK_BF16_TFLOPS
                    roof["isolated_avg_launch_ms"] = iso_cls[dom]["avg_launch_ms"]
                roof["isolated_pass"] = {
                    "note": "same step, micro-batches serialised on one stream, run right after the timed region",
                    "classes": {c: {"tflops": v["tflops"], "frac": v["tflops"] / PEAK_BF16_TFLOPS,
                                    "avg_launch_ms": v["avg_launch_ms"]} for c, v in iso_cls.items()},
                    "breakdown_ms_per_step": {k: v[0] / 2 for k, v in iso.items() if v[1] > 0}}
        cpu, parity, recall, api, tok16, ragged, parity_rel, parity_prob, e2e = None, None, None, None, None, None, None, None, None
        if world == 1 and args.cpu_budget > 0:
            if roof is not None:
                roof.update(power_limited_rate(shape, min(args.micro_batch_tokens or n_chunks * SEQ, n_chunks * SEQ), local_rank, value))
            api = api_leg(eng, shape, n_chunks, steps=max(3, args.steps))
            if api and "api_chunks_per_s" in api:
                api["fraction_of_resident_rate"] = api["api_chunks_per_s"] / value
            ragged = ragged_leg(shape, weights, qa_w, qa_b, n_chunks * SEQ, args.micro_batch_tokens, local_rank, max(3, args.steps), value * SEQ)
            ragged_best = ragged_leg(shape, weights, qa_w, qa_b, n_chunks * SEQ, args.micro_batch_tokens, local_rank, 3, value * SEQ, aligned=True)
            if ragged is not None and ragged_best is not None:
                ragged["aligned_best_case"] = {k: ragged_best.get(k) for k in ("ragged_chunks_per_s", "tokens_per_s_vs_512_token_batch",
                                                                                  "chunks_per_step", "tokens_per_step", "lengths", "error")
                                               if k in ragged_best}
            tok16 = token_head_f16_leg(shape, weights, seqs, args.micro_batch_tokens, local_rank, steps=max(3, args.steps))
            recall = topk_recall_check()
            e2e = e2e_text_leg(shape, weights, qa_w, qa_b, local_rank)
            cpu, ref_logits = cpu_baseline(shape, weights, qa_w, qa_b, seqs, bounds, args.cpu_budget)
            ref = np.concatenate(ref_logits, axis=0)
            got = logits[: ref.shape[0]]
            parity = float(np.abs(got - ref).max())
            # relative to the logit scale of the head (a trained head has a larger one than this random-init head), and in the
            # probability the extractor thresholds (softmax over the two classes, extractors.py:270-277)
            parity_rel = float(np.abs(got - ref).max() / max(float(np.abs(ref).max()), 1e-30))
            parity_prob = float(np.abs(1.0 / (1.0 + np.exp(got[:, 0] - got[:, 1])) - 1.0 / (1.0 + np.exp(ref[:, 0] - ref[:, 1]))).max())
        out = {
            "metric": "query x chunk span-extractions/sec @512tok (ModernBERT-base extractor, chunks/s)",
            "value": value, "unit": "chunks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.operand_dtype, "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: ModernBERT-base span extractor, batch 256 chunks x 512 tok, single query, 16 sentences/chunk" if args.model == "base" else "ModernBERT-large geometry (BASELINE configs[4] extractor), batch 256 chunks x 512 tok, 16 sentences/chunk"),
                       "chunks_per_gpu_per_step": n_chunks, "seq_len": SEQ, "sentences_per_chunk": N_SENT,
                       "micro_batch_tokens": args.micro_batch_tokens, "parallelism": f"dp{world} (independent chunks, no collective)",
                       "weights": f"random-init ModernBERT-{args.model} (seed 1234), {args.operand_dtype} MFMA operands, fp32 accumulate / LayerNorm / softmax / heads; residual stream between sub-layers as the operand copy + one remainder byte per element (16 significant bits with bf16 operands, 19 with fp16)"},
            "sentence_classifications_per_s": value * N_SENT,
            "model_tflops": value * chunk_flops(shape) / 1e12,
            "model_mfma_frac": value * chunk_flops(shape) / 1e12 / (PEAK_BF16_TFLOPS * world),
            "roofline": roof, "cpu_baseline": cpu, "parity_max_abs_err_vs_oracle": parity, "parity_max_rel_err": parity_rel,
            "parity_max_prob_err": parity_prob, "ragged_chunks_per_s": ragged.get("ragged_chunks_per_s") if ragged else None, "ragged_leg": ragged,
            "per_rank_ms_per_step": per_rank_ms, "world_size_reported_by_backend": backend_world,
            "topk_recall_vs_cpu_ref": recall,
            "sharded_queries_per_s": None, "sharded_topk": None,
            "api_chunks_per_s": api.get("api_chunks_per_s") if api else None, "api_leg": api, "token_head_f16": tok16,
            "e2e_queries_per_s": e2e.get("e2e_queries_per_s") if e2e else None, "e2e_text_leg": e2e,
            "breakdown": breakdown,
        }
    else:
        out = None
    eng.close()
    if world > 1:
        # Second timed leg (the retrieval exchange).  It is the first code of this repository to meet a real xGMI ring -- a new RCCL
        # communicator beside torch's, an all-gather on it -- so it runs under a watchdog: if it has not returned after
        # VRAG_BENCH_SHARDED_TIMEOUT seconds (default 240) rank 0 prints the line it already has, with the timeout as the leg's error,
        # and every rank leaves; a hang in the exchange cannot take the headline with it.
        import threading

        done = threading.Event()

        def bail():
            if done.is_set():
                return
            if rank == 0 and out is not None:
                out["sharded_topk"] = {"error": f"sharded retrieval leg did not finish within {limit:.0f} s (watchdog)"}
                print(json.dumps(out), flush=True)
            print(f"bench.py: rank {rank}: sharded retrieval leg timed out; leaving", file=sys.stderr, flush=True)
            os._exit(0)

        limit = float(os.environ.get("VRAG_BENCH_SHARDED_TIMEOUT", "240"))
        timer = threading.Timer(limit, bail)
        timer.daemon = True
        timer.start()
        sharded = sharded_retrieval_leg(rank, world, local_rank, backend)
        done.set()
        timer.cancel()
        if rank == 0:
            out["sharded_queries_per_s"] = sharded.get("sharded_queries_per_s") if sharded else None
            out["sharded_topk"] = sharded
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()   # leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
