#!/usr/bin/env python3
"""Small-batch latency probe: one query's worth of work (k chunks of ~S tokens) through the ModernBERT-base
extractor path, device work only (ids already packed on the host): load_batch + run + qa head + read-back."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    import verbatim_rag_amd  # noqa: F401
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape
    from verbatim_rag_amd.weights import random_init, random_qa_head

    shape = ModernBertShape.base()
    eng = EncoderEngine(shape, random_init(shape, 1234), max_tokens=32768, max_seqs=64, max_seq_len=512, max_ranges=4096)
    eng.set_qa_head(*random_qa_head(shape))
    rng = np.random.default_rng(0)
    for k, S in ((1, 200), (5, 200), (5, 512), (8, 512), (16, 512), (32, 512), (64, 512)):
        seqs = [rng.integers(1000, 50000, size=S).astype(np.int32) for _ in range(k)]
        bounds = [[(1 + 12 * j, 12 * j + 11) for j in range(S // 12 - 1)] for _ in range(k)]
        for _ in range(5):
            eng.qa_logits(seqs, bounds)
        t0 = time.perf_counter()
        for _ in range(args.iters):
            eng.qa_logits(seqs, bounds)
        dt = (time.perf_counter() - t0) / args.iters
        print(json.dumps({"chunks": k, "tokens_per_chunk": S, "ms_per_call": dt * 1e3, "calls_per_s": 1 / dt,
                          "chunks_per_s": k / dt}))
    eng.close()


if __name__ == "__main__":
    main()
