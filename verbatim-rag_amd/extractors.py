"""GPU drop-in for the reference's ModelSpanExtractor.

Plug point kept: `SpanExtractor.extract_spans(question, search_results) -> Dict[text, List[str]]`
(packages/core/verbatim_core/extractors.py:34-54); callers: verbatim_rag/core.py:255,351,
verbatim_core/transform.py:94,122, verbatim_rag/streaming.py:98-100.

What changes underneath: the reference runs one tokenisation + one [1,S] ModernBERT forward per
retrieved chunk (extractors.py:233-268); here all (question, chunk) pairs of a call are packed
into ONE padding-free token batch and run through the HIP encoder + sentence head in one go.
The host-side rules are the reference's, unchanged: regex sentence split (:190-195), the
`[CLS] q [SEP] s1 [SEP] ...` packer with the 510-token budget (dataset.py:109-243),
`softmax(logits)[:,1] > threshold` (strict) selection (:270-277), dict keyed by chunk text.
"""
from __future__ import annotations

import json
import logging
import os
import threading
import time
from abc import ABC, abstractmethod
from concurrent.futures import Future
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .packing import (
    PackedSample,
    TokenizerAdapter,
    encode_question_and_sentences,
    split_into_sentences,
    split_into_sentences_batch,
    valid_boundaries,
)

logger = logging.getLogger(__name__)


class SpanExtractor(ABC):
    """Same abstract interface as the reference (extractors.py:34-54)."""

    @abstractmethod
    def extract_spans(self, question: str, search_results: List[Any]) -> Dict[str, List[str]]:
        raise NotImplementedError

    async def extract_spans_async(self, question: str, search_results: List[Any]) -> Dict[str, List[str]]:
        import asyncio

        return await asyncio.to_thread(self.extract_spans, question, search_results)


def softmax_rows(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.float32)
    x = x - x.max(axis=-1, keepdims=True)
    e = np.exp(x, dtype=np.float32)
    return (e / e.sum(axis=-1, keepdims=True, dtype=np.float32)).astype(np.float32)


def select_sentences(logits: np.ndarray, raw_sentences: Sequence[str], threshold: float) -> List[str]:
    """extractors.py:270-277: strict `>`, logits row i is matched with raw_sentences[i]."""
    spans: List[str] = []
    if logits is None or len(logits) == 0:
        return spans
    probs = softmax_rows(np.asarray(logits))
    for i in range(probs.shape[0]):
        if i < len(raw_sentences) and probs[i, 1] > threshold:
            spans.append(raw_sentences[i])
    return spans


def token_spans_to_char_spans(
    probs: Sequence[float],
    offsets: Sequence[Tuple[int, int]],
    context: str,
    threshold: float,
    min_span_chars: int,
    merge_gap_chars: int,
) -> List[str]:
    """Highlighter post-processing (the hub `.process()` source is not part of the reference;
    knobs from extractors.py:86-113, run/merge semantics after `_find_span_regions`,
    extractors.py:438-469, applied in character space as the knob names say):
    runs of tokens with P > threshold -> [char_start, char_end); merge runs whose gap is
    <= merge_gap_chars; drop spans shorter than min_span_chars; spans are exact substrings."""
    runs: List[List[int]] = []
    cur: Optional[List[int]] = None
    for p, (a, b) in zip(probs, offsets):
        if b <= a:  # special / empty-offset token
            continue
        if p > threshold:
            if cur is None:
                cur = [a, b]
            else:
                cur[1] = max(cur[1], b)
        elif cur is not None:
            runs.append(cur)
            cur = None
    if cur is not None:
        runs.append(cur)
    merged: List[List[int]] = []
    for r in runs:
        if merged and r[0] - merged[-1][1] <= merge_gap_chars:
            merged[-1][1] = max(merged[-1][1], r[1])
        else:
            merged.append(list(r))
    out = []
    for a, b in merged:
        if b - a >= min_span_chars and context[a:b].strip():
            out.append(context[a:b])
    return out


class GpuModelSpanExtractor(SpanExtractor):
    """MI355X implementation of ModelSpanExtractor (extractors.py:57-279).

    Construct either from a local HF checkpoint directory (`model_path`: config.json +
    *.safetensors [+ tokenizer.json]) or from an already-built `engine` + `tokenizer`.
    Raises at construction when the HIP library or a GPU is missing -- there is no CPU path.
    """

    DEFAULT_MODEL = "KRLabsOrg/verbatim-rag-modern-bert-v2"
    GPU_SPLIT_MIN = 256   # unseen chunks in one call from which the sentence split runs on the GPU (prepare_chunks at ingest)
    _FORMAT_HIGHLIGHTER = "highlighter"
    _FORMAT_QA_MODEL = "qa_model"

    def __init__(
        self,
        model_path: Optional[str] = None,
        device: int | str | None = None,
        threshold: float = 0.2,
        extraction_mode: str = "individual",
        max_display_spans: int = 5,
        min_span_chars: int = 30,
        merge_gap_chars: int = 20,
        max_length: int = 8192,
        doc_stride: int = 256,
        *,
        engine: Any = None,
        tokenizer: Any = None,
        model_format: Optional[str] = None,
        qa_max_length: int = 512,
        max_batch_tokens: int = 65536,
        max_batch_seqs: int = 512,
        chunk_cache_size: int = 65536,
        extra_engines: Sequence[Any] = (),
        n_engines: int = 1,
        operand_dtype: Optional[str] = None,
    ):
        """operand_dtype: MFMA operand type of the engines this extractor builds (`EncoderEngine`): None picks "bf16" for
        the sentence-classifier format (sentence logits within 3e-4 of the fp32 reference) and "f16" for the v2 highlighter,
        whose per-token logits need the three extra mantissa bits to stay within 1e-3."""
        self.operand_dtype = operand_dtype
        self.model_path = model_path
        self.threshold = threshold
        self.min_span_chars = min_span_chars
        self.merge_gap_chars = merge_gap_chars
        self.max_length = max_length
        self.doc_stride = doc_stride
        self.qa_max_length = qa_max_length
        self.max_batch_tokens = max_batch_tokens
        self.max_batch_seqs = max_batch_seqs
        self._lock = threading.Lock()  # callers arrive from asyncio.to_thread workers (extractors.py:54)
        # chunk text -> (sentences, per-sentence token ids): chunk texts are known at ingest and recur across
        # queries, and the reference's per-query tokenisation (2.7 ms/chunk) would cap the GPU path
        self._chunk_cache: Dict[str, tuple] = {}   # text -> _cache_entry
        self._chunk_cache_size = chunk_cache_size
        self._cache_lock = threading.Lock()
        self._pool = None
        dev = 0 if device in (None, "cuda", "cpu", "mps") else int(str(device).replace("cuda:", ""))
        self.device = f"cuda:{dev}"

        if engine is not None:
            if tokenizer is None:
                raise ValueError("engine= needs tokenizer=")
            self.engine = engine
            self._format = model_format or (
                self._FORMAT_HIGHLIGHTER if getattr(engine, "token_labels", 0) and not getattr(engine, "qa_labels", 0)
                else self._FORMAT_QA_MODEL)
        else:
            if model_path is None or not os.path.isdir(model_path):
                raise FileNotFoundError(
                    f"model_path={model_path!r}: a local HF checkpoint directory is required (no network here); "
                    "or pass engine= and tokenizer=")
            self._format = model_format or self._detect_format(model_path)
            self.engine = self._build_engine(model_path, dev)
            tokenizer = tokenizer or self._load_tokenizer(model_path)
            extra_engines = [self._build_engine(model_path, dev) for _ in range(max(1, int(n_engines)) - 1)]
        # Further handles of the same model (own weights copy + workspace + streams): a multi-sub-batch call alternates
        # between them from worker threads, so one handle's upload / read-back / host turnaround hides behind the
        # other's kernels (measured 0.42 -> 0.33 s for 5000 pairs, DESIGN.md "Serving shape").
        self.engines = [self.engine] + list(extra_engines)
        self._locks = [getattr(e, "lock", None) or threading.Lock() for e in self.engines]   # a handle's own lock: wrappers may share it
        self._lock = self._locks[0]
        self.tokenizer = tokenizer
        self._tok = TokenizerAdapter.for_model(tokenizer, self.engine.shape)
        logger.info("GpuModelSpanExtractor ready: format=%s device=%s", self._format, self.device)

    # ------------------------------------------------------------------ loading
    @staticmethod
    def _detect_format(model_path: str) -> str:
        """extractors.py:135-149 without transformers: auto_map naming a *Highlighter* class."""
        try:
            with open(os.path.join(model_path, "config.json")) as f:
                cfg = json.load(f)
            auto_map = cfg.get("auto_map") or {}
            target = auto_map.get("AutoModel") or auto_map.get("AutoModelForTokenClassification")
            if target and "Highlighter" in target:
                return GpuModelSpanExtractor._FORMAT_HIGHLIGHTER
        except Exception as exc:
            logger.warning("Highlighter detection failed for %s: %s", model_path, exc)
        return GpuModelSpanExtractor._FORMAT_QA_MODEL

    def _build_engine(self, model_path: str, dev: int):
        from .engine import EncoderEngine
        from .weights import load_safetensors_dir

        shape, tensors, _cfg = load_safetensors_dir(model_path)
        max_seq = self.qa_max_length if self._format == self._FORMAT_QA_MODEL else self.max_length
        dtype = self.operand_dtype or ("bf16" if self._format == self._FORMAT_QA_MODEL else "f16")
        eng = EncoderEngine(shape, tensors, max_tokens=self.max_batch_tokens, max_seqs=self.max_batch_seqs,
                            max_seq_len=max_seq, max_ranges=max(4096, self.max_batch_tokens // 8), device=dev,
                            operand_dtype=dtype)
        if self._format == self._FORMAT_QA_MODEL:
            eng.set_qa_head(tensors["classifier.weight"], tensors["classifier.bias"])
        else:
            eng.set_token_head(tensors["head.dense.weight"], tensors["head.norm.weight"],
                               tensors["classifier.weight"], tensors["classifier.bias"])
        return eng

    def _f16_clamped(self, engine) -> bool:
        """fp16 operands saturate at 65504 instead of overflowing; a checkpoint with activation outliers beyond that
        would come back with plausible, wrong logits.  The library reports every clamp (`vrag_encoder_f16_saturated`):
        on the first report the engines this extractor built are rebuilt with bf16 operands (fp32's exponent range)
        and True is returned -- the caller runs its batch again; an extractor that was handed its engine cannot rebuild
        it and raises (its per-chunk error handling logs the failure and returns no spans: never silently wrong)."""
        if getattr(engine, "operand_dtype", "bf16") != "f16" or not hasattr(engine, "f16_saturated") or not engine.f16_saturated(reset=True):
            return False
        if self.model_path is None or not os.path.isdir(self.model_path):
            raise RuntimeError("fp16 MFMA operands saturated on this checkpoint (activations beyond 65504): "
                               "build the engine with operand_dtype='bf16'")
        with self._cache_lock:
            if self.operand_dtype != "bf16":               # the first thread to notice rebuilds; the others just retry
                logger.warning("fp16 MFMA operands saturated on %s (activations beyond 65504): switching this extractor to "
                               "bf16 operands (construct it with operand_dtype='bf16' to skip the probe)", self.model_path)
                dev = int(self.device.replace("cuda:", ""))
                self.operand_dtype = "bf16"
                # swapped in as a whole; a call still running on an old handle keeps it alive and finishes on it
                self.engines = [self._build_engine(self.model_path, dev) for _ in self.engines]
                self.engine = self.engines[0]
        return True

    @staticmethod
    def _load_tokenizer(model_path: str):
        tj = os.path.join(model_path, "tokenizer.json")
        try:
            from transformers import AutoTokenizer

            return AutoTokenizer.from_pretrained(model_path)
        except Exception as e:  # fall back to the raw tokenizers file
            logger.warning("AutoTokenizer failed for %s (%s); using tokenizers.Tokenizer", model_path, e)
            from tokenizers import Tokenizer

            return Tokenizer.from_file(tj)

    def _split_into_sentences(self, text: str) -> List[str]:
        return split_into_sentences(text)

    # ------------------------------------------------------------------ API
    def extract_spans(self, question: str, search_results: List[Any]) -> Dict[str, List[str]]:
        if self._format == self._FORMAT_HIGHLIGHTER:
            return self._extract_highlighter(question, search_results)
        return self._extract_qa_model(question, search_results)

    # ------------------------------------------------------------------ legacy qa_model path
    def prepare_chunks(self, texts: Sequence[str]) -> None:
        """Optional ingest-time hook: pre-split and pre-tokenise chunk texts (e.g. from add_vectors)."""
        self.pack_qa("", list(texts))

    def pack_qa(self, question: str, texts: Sequence[str]) -> Tuple[List[List[str]], List[Optional[PackedSample]]]:
        """Sentence split + token packing for every chunk; one batched tokenizer call for all
        sentences (bit-identical to the reference's per-sentence calls, dataset.py:158-167)."""
        budget = self.qa_max_length - 2
        q_ids = self._tok.ids(question, add_special_tokens=True, max_length=budget)
        all_sents, samples = [], []
        for sents, ids, *_rest in self._entries(texts):
            all_sents.append(sents)
            if not sents:
                samples.append(None)
                continue
            samples.append(encode_question_and_sentences(q_ids, ids, self._tok.sep_token_id, max_length=self.qa_max_length))
        return all_sents, samples

    def _entries(self, texts: Sequence[str]):
        """Sentence split + tokenisation per chunk, memoised (sentences are tokenised independently of the
        question, dataset.py:158-167, so the cached ids are exactly what the reference would produce); one
        batched tokenizer call for every sentence of every chunk not seen before."""
        with self._cache_lock:
            missing = [t for t in dict.fromkeys(texts) if t not in self._chunk_cache]
            if missing:
                # ingest-sized batches of unseen chunks: boundaries on the GPU (one lane per chunk); a query's few: the regex
                split = split_into_sentences_batch(missing, int(self.device.replace("cuda:", ""))) if len(missing) >= self.GPU_SPLIT_MIN \
                    else [split_into_sentences(t) for t in missing]
                flat = [s for sents in split for s in sents]
                flat_ids = self._tok.ids_batch(flat, max_length=self.qa_max_length - 2)
                if len(self._chunk_cache) + len(missing) > self._chunk_cache_size:
                    self._chunk_cache.clear()
                o = 0
                for t, sents in zip(missing, split):
                    self._chunk_cache[t] = self._cache_entry(sents, flat_ids[o:o + len(sents)])
                    o += len(sents)
            return [self._chunk_cache[t] for t in texts]

    def _cache_entry(self, sents: List[str], ids: List[List[int]]):
        """(sentences, per-sentence ids, tail, cum, no sentence is empty): tail = [SEP] s1 [SEP] s2 ... as one int32 array, cum[i] = tokens of
        the first i+1 `[SEP] sentence` groups -- the question-independent part of the packer, computed once per chunk."""
        sep = self._tok.sep_token_id
        tail = np.fromiter((x for sent in ids for x in [sep, *sent]), dtype=np.int32) if ids else np.zeros(0, np.int32)
        cum = np.cumsum([len(sent) + 1 for sent in ids], dtype=np.int64) if ids else np.zeros(0, np.int64)
        # the addresses of tail / cum ride along for the native packer (vrag_pack_qa_pairs): the arrays live in this tuple
        return (sents, ids, tail, cum, all(len(sent) > 0 for sent in ids), tail.ctypes.data, cum.ctypes.data)

    def _pack_fast(self, q_ids: List[int], entry):
        """The packer (packing.encode_question_and_sentences, dataset.py:127-243) on the cached pieces: the sentences
        that fit are a prefix (the reference stops at the first one that does not), so the cut is one searchsorted.
        Returns (input ids, inclusive starts, inclusive ends) or None when the general routine must decide (the
        question alone reaches the budget, or a sentence without tokens makes an empty range)."""
        sents, ids, tail, cum, no_empty = entry[:5]
        budget = self.qa_max_length - 2
        q = q_ids[:-1] if len(q_ids) > 1 and q_ids[-1] == self._tok.sep_token_id else q_ids
        qlen = len(q)
        if qlen >= budget or len(cum) == 0 or not no_empty:
            return None
        m = int(np.searchsorted(cum, budget - qlen, side="right"))      # groups with qlen + cum[i] <= budget
        if m == 0:
            return None
        ends = qlen + cum[:m] - 1
        starts = np.empty(m, np.int64)
        starts[0] = qlen + 1
        starts[1:] = qlen + cum[: m - 1] + 1
        n = qlen + int(cum[m - 1])
        out = np.empty(n + (1 if n < budget else 0), np.int32)
        out[:qlen] = q
        out[qlen:n] = tail[: int(cum[m - 1])]
        if n < budget:
            out[n] = self._tok.sep_token_id                              # dataset.py:202-205
        if m < len(cum):
            logger.warning("Legacy QA input exceeded the %d-token budget; dropping %d sentence(s)", budget + 2, len(cum) - m)
        return out, starts, ends

    def _pack_fast_many(self, q_ids: List[int], entries):
        """`_pack_fast` for every chunk of ONE question in one native call (`vrag_pack_qa_pairs`, host code of the C ABI): element
        i is what `_pack_fast(q_ids, entries[i])` returns -- views into three flat arrays instead of a dozen small numpy
        operations per chunk (3-4 ms of interpreter time per 256-chunk call: a tenth of the device time at 512 tokens)."""
        import ctypes as C

        from . import _lib

        lib = _lib.load()
        if lib is None:   # the CPU test suites stub the library out: the same packer, chunk by chunk, in numpy
            return [self._pack_fast(q_ids, e) if e[0] else None for e in entries]
        res = [None] * len(entries)
        budget = self.qa_max_length - 2
        q = q_ids[:-1] if len(q_ids) > 1 and q_ids[-1] == self._tok.sep_token_id else q_ids
        qlen = len(q)
        idx = [i for i, e in enumerate(entries) if e[0] and e[4] and len(e[3])] if qlen < budget else []
        if not idx:
            return res
        n = len(idx)
        ng = [len(entries[i][3]) for i in idx]
        tails = (C.c_uint64 * n)(*[entries[i][5] for i in idx])
        cums = (C.c_uint64 * n)(*[entries[i][6] for i in idx])
        ng_a = np.asarray(ng, np.int32)
        q_a = np.asarray(q, np.int32) if qlen else np.zeros(1, np.int32)
        ids_out = np.empty(n * budget, np.int32)
        n_rng = int(ng_a.sum())
        starts, ends = np.empty(n_rng, np.int64), np.empty(n_rng, np.int64)
        seq_lens, kept, totals = np.empty(n, np.int32), np.empty(n, np.int32), np.zeros(2, np.int64)
        ip, lp = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        _lib.check("vrag_pack_qa_pairs", lib.vrag_pack_qa_pairs(
            q_a.ctypes.data_as(ip), qlen, n, tails, cums, ng_a.ctypes.data_as(ip), budget, self._tok.sep_token_id,
            ids_out.ctypes.data_as(ip), ids_out.size, starts.ctypes.data_as(lp), ends.ctypes.data_as(lp), n_rng,
            seq_lens.ctypes.data_as(ip), kept.ctypes.data_as(ip), totals.ctypes.data_as(lp)))
        io = ro = 0
        for i, m, ln, g in zip(idx, kept.tolist(), seq_lens.tolist(), ng):
            if m == 0:
                continue                                   # nothing fits: the general routine decides (and logs)
            res[i] = (ids_out[io:io + ln], starts[ro:ro + m], ends[ro:ro + m])
            io += ln
            ro += m
            if m < g:
                logger.warning("Legacy QA input exceeded the %d-token budget; dropping %d sentence(s)", budget + 2, g - m)
        return res

    def _extract_qa_model(self, question: str, search_results: List[Any]) -> Dict[str, List[str]]:
        return self.extract_spans_batch([question], [search_results])[0]

    def extract_spans_batch(self, questions: Sequence[str],
                            results_per_question: Sequence[Sequence[Any]]) -> List[Dict[str, List[str]]]:
        """Cross-query batching: all (question_i, chunk_ij) pairs of several concurrent
        queries go through the GPU as shared padding-free batches.  Element i of the result equals
        `extract_spans(questions[i], results_per_question[i])` (both model formats)."""
        if self._format != self._FORMAT_QA_MODEL:
            return self._extract_highlighter_batch(questions, results_per_question)
        out: List[Dict[str, List[str]]] = []
        budget = self.qa_max_length - 2
        cur: list = []   # (query index, text, sentences, ids int32[], starts int64[], ends int64[])
        tok = rng = 0
        pending = []     # sub-batches in flight on the worker thread while this thread packs the next one

        def add(item):
            nonlocal cur, tok, rng
            n_tok, n_rng = len(item[3]), len(item[4])
            if n_tok > self.engine.max_tokens or n_rng > self.engine.max_ranges:
                # this chunk stays without spans, the rest of the call goes on (extractors.py:225-227: log, [] for the chunk)
                logger.error("query %d: a sample of %d tokens / %d sentences exceeds the engine workspace", item[0], n_tok, n_rng)
                return
            if len(cur) >= self.engine.max_seqs or tok + n_tok > self.engine.max_tokens or rng + n_rng > self.engine.max_ranges:
                pending.append(self._worker().submit(self._run_sub_batch, cur, out, len(pending) % len(self.engines)))
                cur, tok, rng = [], 0, 0
            cur.append(item)
            tok += n_tok
            rng += n_rng

        try:
            for qi, (question, results) in enumerate(zip(questions, results_per_question)):
                texts = [getattr(r, "text", "") for r in results]
                out.append({t: [] for t in texts})
                q_ids = self._tok.ids(question, add_special_tokens=True, max_length=budget)
                entries = self._entries(texts)
                fasts = self._pack_fast_many(q_ids, entries)
                for i, (t, entry) in enumerate(zip(texts, entries)):
                    if not entry[0]:
                        continue                              # blank chunk -> [] (extractors.py:209-211)
                    fast = fasts[i]
                    if fast is not None:
                        add((qi, t, entry[0], fast[0], fast[1], fast[2]))
                        continue
                    smp = encode_question_and_sentences(q_ids, entry[1], self._tok.sep_token_id, max_length=self.qa_max_length)
                    vb = valid_boundaries(smp.sentence_boundaries, len(smp.input_ids))
                    if not vb:
                        # the reference's QAModel returns None here and `len(None)` raises; we log and return [].
                        logger.error("query %d chunk %d: no sentence fits the %d-token budget", qi, i, self.qa_max_length)
                        continue
                    add((qi, t, entry[0], np.asarray(smp.input_ids, np.int32), np.asarray([b[0] for b in vb], np.int64),
                         np.asarray([b[1] for b in vb], np.int64)))
            if cur:
                if pending:
                    pending.append(self._worker().submit(self._run_sub_batch, cur, out, len(pending) % len(self.engines)))
                else:
                    self._run_sub_batch(cur, out)             # the common single-query call: no thread hop
        finally:
            for f in pending:                                 # never leave a sub-batch running behind the caller's back
                f.result()
        return out

    def _worker(self):
        """One thread per engine handle drives the device for multi-sub-batch calls: the C call releases the GIL, so
        the caller packs sub-batch n+1 while sub-batch n is on the GPU."""
        with self._cache_lock:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor

                self._pool = ThreadPoolExecutor(max_workers=len(self.engines), thread_name_prefix="vrag-extract")
            return self._pool

    def _run_sub_batch(self, batch, out, which: int = 0) -> None:
        """One workspace-sized batch: device logits, softmax, strict `>` threshold (extractors.py:272-275)."""
        engine = self.engines[which]
        with self._locks[which]:
            try:
                counts = np.asarray([len(b[4]) for b in batch], np.int64)
                if hasattr(engine, "qa_logits_packed"):
                    flat = engine.qa_logits_packed(
                        np.concatenate([b[3] for b in batch]), np.asarray([len(b[3]) for b in batch], np.int32),
                        np.repeat(np.arange(len(batch), dtype=np.int32), counts),
                        np.concatenate([b[4] for b in batch]), np.concatenate([b[5] for b in batch]))
                else:   # engines without the flat entry point
                    flat = np.concatenate(engine.qa_logits(
                        [b[3] for b in batch], [list(zip(b[4].tolist(), b[5].tolist())) for b in batch]))
                if self._f16_clamped(engine):
                    return self._run_sub_batch(batch, out, which)          # once more, on the bf16 engines
                keep = softmax_rows(flat)[:, 1] > self.threshold
                # selected sentences chunk by chunk, from ONE nonzero over the batch's ranges (every chunk starts as [])
                sel = np.nonzero(keep)[0]
                ends = np.cumsum(counts)
                owner = np.searchsorted(ends, sel, side="right")
                local = sel - (ends - counts)[owner]
                picked: Dict[int, List[int]] = {}
                for b, i in zip(owner.tolist(), local.tolist()):
                    picked.setdefault(b, []).append(i)
                for b, (qi, text, sents, _ids, _st, _en) in enumerate(batch):   # in batch order: a chunk listed twice keeps its last evaluation
                    idx = picked.get(b)
                    out[qi][text] = [sents[i] for i in idx if i < len(sents)] if idx else []
            except Exception as exc:  # same contract as extractors.py:225-227: log, [] for the chunk(s)
                logger.error("GPU span extraction failed: %s", exc)

    # ------------------------------------------------------------------ v2 highlighter path
    def _encode_windows(self, question: str, context: str):
        """Pair-encode (question, context) into windows of <= max_length tokens overlapping by
        doc_stride context tokens.  Returns [(ids, ctx_token_slice, first_ctx_pos_in_window)], offsets."""
        tok = self.tokenizer
        if hasattr(tok, "encode_batch") and tok.__class__.__module__.startswith("tokenizers"):
            enc = tok.encode(context, add_special_tokens=False)
            ctx_ids, offsets = list(enc.ids), list(enc.offsets)
        else:
            enc = tok(context, add_special_tokens=False, return_offsets_mapping=True)
            ctx_ids, offsets = list(enc["input_ids"]), [tuple(o) for o in enc["offset_mapping"]]
        q = self._tok.ids(question, add_special_tokens=True, max_length=max(8, self.max_length // 2))
        room = self.max_length - len(q) - 1
        if room <= 0:
            raise ValueError("question leaves no room for context tokens")
        step = max(1, room - self.doc_stride)
        windows = []
        a = 0
        while True:
            b = min(len(ctx_ids), a + room)
            ids = q + ctx_ids[a:b] + [self._tok.sep_token_id]
            windows.append((ids, (a, b), len(q)))
            if b >= len(ctx_ids):
                break
            a += step
        return windows, offsets, len(ctx_ids)

    def _extract_highlighter(self, question: str, search_results: List[Any]) -> Dict[str, List[str]]:
        return self._extract_highlighter_batch([question], [search_results])[0]

    def _extract_highlighter_batch(self, questions: Sequence[str],
                                   results_per_question: Sequence[Sequence[Any]]) -> List[Dict[str, List[str]]]:
        """v2 path for several queries at once: the windows of every (question, chunk) pair share padding-free GPU
        batches (windows never see each other, so element i equals the single-query call)."""
        out: List[Dict[str, List[str]]] = []
        jobs = []   # (query index, context, windows, offsets, context tokens)
        for qi, (question, search_results) in enumerate(zip(questions, results_per_question)):
            relevant: Dict[str, List[str]] = {}
            out.append(relevant)
            for result in search_results:
                context = getattr(result, "text", "")
                relevant[context] = []
                if not context.strip():
                    continue
                try:
                    windows, offsets, n_ctx = self._encode_windows(question, context)
                    jobs.append((qi, context, windows, offsets, n_ctx))
                except Exception as exc:
                    logger.error("Highlighter extraction failed: %s", exc)
        flat = [(ji, w) for ji, job in enumerate(jobs) for w in job[2]]
        probs = [np.zeros(job[4], dtype=np.float32) for job in jobs]
        with self._lock:
            start = 0
            while start < len(flat):
                tok = 0
                end = start
                while end < len(flat) and end - start < self.engine.max_seqs and \
                        tok + len(flat[end][1][0]) <= self.engine.max_tokens:
                    tok += len(flat[end][1][0])
                    end += 1
                if end == start:
                    # one window larger than the workspace: this chunk stays without spans, the others go on (the
                    # reference logs a failing chunk and returns [] for it, extractors.py:225-227)
                    logger.error("Highlighter extraction failed: a window of %d tokens exceeds the engine workspace "
                                 "(max_tokens=%d)", len(flat[start][1][0]), self.engine.max_tokens)
                    start += 1
                    continue
                try:
                    self.engine.load_batch([w[0] for _ji, w in flat[start:end]])
                    self.engine.run()
                    self.engine.run_token_head()
                    logits = self.engine.read_token_logits()
                    if self._f16_clamped(self.engine):
                        continue                                # the engines run on bf16 operands now: this batch again
                    p1 = softmax_rows(logits)[:, 1]
                    o = 0
                    for ji, (ids, (a, b), q_len) in flat[start:end]:
                        seg = p1[o + q_len:o + q_len + (b - a)]
                        probs[ji][a:b] = np.maximum(probs[ji][a:b], seg)
                        o += len(ids)
                except Exception as exc:
                    logger.error("Highlighter extraction failed: %s", exc)
                start = end
        for (qi, context, _w, offsets, _n), p in zip(jobs, probs):
            out[qi][context] = token_spans_to_char_spans(p, offsets, context, self.threshold, self.min_span_chars,
                                                         self.merge_gap_chars)
        return out


class CoalescingSpanExtractor(SpanExtractor):
    """Host scheduler in front of an extractor that offers `extract_spans_batch`.

    The reference calls `extract_spans` once per query, from `asyncio.to_thread` workers when it serves
    concurrent requests (extractors.py:48-54, streaming.py:98-100); each call is a handful of chunks
    (k = 5), far too little to fill the GPU.  This wrapper keeps the reference's per-query interface and
    coalesces calls that arrive within `max_wait_ms` of each other (or until `max_pairs` (question, chunk)
    pairs are waiting) into one `extract_spans_batch` call on a background thread.  Results are identical to
    per-call extraction (cross-query batching is bit-neutral: packed sequences never see each other).
    A failing batch is retried query by query so one bad request cannot fail its neighbours.
    """

    def __init__(self, inner: Any, max_wait_ms: float = 2.0, max_pairs: int = 512):
        if not hasattr(inner, "extract_spans_batch"):
            raise TypeError("inner extractor must provide extract_spans_batch(questions, results_per_question)")
        self.inner = inner
        self.max_wait = max(0.0, float(max_wait_ms)) / 1e3
        self.max_pairs = int(max_pairs)
        self._cv = threading.Condition()
        self._queue: List[Tuple[str, List[Any], "Future"]] = []
        self._closed = False
        self.batches_run = 0          # observability: batches issued / queries served
        self.queries_served = 0
        self._thread = threading.Thread(target=self._loop, name="vrag-coalescer", daemon=True)
        self._thread.start()

    # ------------------------------------------------------------------ SpanExtractor API
    def extract_spans(self, question: str, search_results: List[Any]) -> Dict[str, List[str]]:
        fut: Future = Future()
        with self._cv:
            if self._closed:
                raise RuntimeError("CoalescingSpanExtractor is closed")
            self._queue.append((question, list(search_results), fut))
            self._cv.notify_all()
        return fut.result()

    def close(self) -> None:
        with self._cv:
            self._closed = True
            self._cv.notify_all()
        self._thread.join(timeout=5.0)

    # ------------------------------------------------------------------ scheduler thread
    def _pending_pairs(self) -> int:
        return sum(len(r) for _q, r, _f in self._queue)

    def _loop(self) -> None:
        while True:
            with self._cv:
                while not self._queue and not self._closed:
                    self._cv.wait()
                if not self._queue and self._closed:
                    return
                # linger: give concurrent callers `max_wait` to join the batch
                deadline = time.monotonic() + self.max_wait
                while self._pending_pairs() < self.max_pairs and not self._closed:
                    left = deadline - time.monotonic()
                    if left <= 0:
                        break
                    self._cv.wait(timeout=left)
                batch, pairs = [], 0
                while self._queue and (not batch or pairs + len(self._queue[0][1]) <= self.max_pairs):
                    item = self._queue.pop(0)
                    batch.append(item)
                    pairs += len(item[1])
            self._run(batch)

    def _run(self, batch) -> None:
        self.batches_run += 1
        try:
            outs = self.inner.extract_spans_batch([b[0] for b in batch], [b[1] for b in batch])
            if len(outs) != len(batch):
                raise RuntimeError("extract_spans_batch returned %d results for %d queries" % (len(outs), len(batch)))
            for (_q, _r, fut), out in zip(batch, outs):
                self.queries_served += 1
                fut.set_result(out)
        except Exception as exc:
            logger.error("coalesced extraction failed (%s); retrying per query", exc)
            for q, r, fut in batch:
                if fut.done():
                    continue
                try:
                    fut.set_result(self.inner.extract_spans(q, r))
                except Exception as exc2:
                    fut.set_exception(exc2)
                self.queries_served += 1
