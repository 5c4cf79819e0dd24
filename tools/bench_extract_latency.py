#!/usr/bin/env python3
"""Host-inclusive latency of the reference-shaped call: `extract_spans(question, 5 results)` on the GPU extractor
(ModernBERT-base shape, random-init), chunk tokenisation cached like after ingest.  Prints one JSON object."""
import json
import os
import sys
import time
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_e2e import make_texts  # noqa: E402


def main():
    from tokenizers import Tokenizer

    import verbatim_rag_amd  # noqa: F401
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape
    from verbatim_rag_amd.extractors import GpuModelSpanExtractor
    from verbatim_rag_amd.weights import random_init, random_qa_head

    rng = np.random.default_rng(3)
    pool = make_texts(rng, 256)
    tok = Tokenizer.from_file(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "tokenizer.json"))
    shape = ModernBertShape.base()
    eng = EncoderEngine(shape, random_init(shape, 1234), max_tokens=16384, max_seqs=64, max_seq_len=512, max_ranges=2048)
    eng.set_qa_head(*random_qa_head(shape))
    ext = GpuModelSpanExtractor(engine=eng, tokenizer=tok, threshold=0.5)
    ext.prepare_chunks(pool)
    q = "Where is the tall iron tower in the city?"
    calls = [[types.SimpleNamespace(text=pool[int(i)]) for i in rng.integers(0, len(pool), 5)] for _ in range(200)]
    def timed():
        for c in calls[:20]:
            ext.extract_spans(q, c)
        t0 = time.perf_counter()
        for c in calls:
            ext.extract_spans(q, c)
        return (time.perf_counter() - t0) / len(calls)

    eng.graph_stats(enable=0)
    dt_eager = timed()                       # every kernel launched by itself (round 2's path)
    eng.graph_stats(enable=8192)
    dt = timed()                             # the layer schedule replayed from a captured HIP graph
    replays, cached = eng.graph_stats()
    # split: host packing vs device call
    t0 = time.perf_counter()
    for c in calls:
        ext.pack_qa(q, [r.text for r in c])
    dpack = (time.perf_counter() - t0) / len(calls)
    print(json.dumps({"call": "extract_spans(question, 5 chunks of ~12 sentences / ~190 tokens)", "ms_per_call": dt * 1e3,
                      "calls_per_s": 1 / dt, "ms_per_call_eager_launches": dt_eager * 1e3, "graph_replays": replays,
                      "graphs_cached": cached, "host_pack_ms": dpack * 1e3}))
    eng.close()


if __name__ == "__main__":
    main()
