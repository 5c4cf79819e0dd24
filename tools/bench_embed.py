#!/usr/bin/env python3
"""Embedding-side probe: BERT-base / DistilBERT-base shaped encoders (random-init, seeded) on
synthetic token batches resident in HBM: texts/s for dense pooling (encoder only) and SPLADE rows (encoder +
MLM head + max-pool).  Parity of the same path at this width is tests/test_bert_gpu.py (tools never touch oracle/).
Prints one JSON object per model."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--texts", type=int, default=256)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()

    import torch

    import verbatim_rag_amd  # noqa: F401
    from verbatim_rag_amd.engine import BertEncoderEngine, BertShape
    from verbatim_rag_amd.weights import random_init_bert

    for shape in (BertShape.bert_base(), BertShape.distilbert_base()):
        W = random_init_bert(shape, seed=1234)
        n, S = args.texts, args.seq
        eng = BertEncoderEngine(shape, W, max_tokens=n * S, max_seqs=n, max_seq_len=S, max_ranges=n,
                                micro_batch_tokens=65536)
        rng = np.random.default_rng(1)
        seqs = [rng.integers(1000, shape.vocab_size, size=S).astype(np.int32) for _ in range(n)]
        eng.load_batch(seqs)
        eng.load_ranges(list(range(n)), [0] * n, [S - 1] * n)
        torch.cuda.synchronize()

        def timed(fn):
            fn(); fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / args.steps

        def dense():
            eng.run()
            eng.run_pool(True)

        def splade():
            eng.run()
            eng.run_splade()

        td, ts = timed(dense), timed(splade)
        H, I, L, V = shape.hidden_size, shape.intermediate_size, shape.num_hidden_layers, shape.vocab_size
        enc_flop = L * (2.0 * S * (4 * H * H + 2 * H * I) + 4.0 * S * S * H)
        head_flop = 2.0 * S * H * H + 2.0 * S * H * V
        out = {"model": f"{shape.model_type}-base shape (L={L}, H={H}, I={I}, V={V}), random-init",
               "batch": f"{n} x {S} tokens", "dense_texts_per_s": n / td, "dense_ms": td * 1e3,
               "dense_tflops": n * enc_flop / td / 1e12, "splade_texts_per_s": n / ts, "splade_ms": ts * 1e3,
               "splade_tflops": n * (enc_flop + head_flop) / ts / 1e12}
        eng.close()
        # the opt-out head (plain 16-bit operands: a third of the decoder work, weights within ~1e-2 instead of 2e-5)
        Wp = {k_: v for k_, v in W.items() if not k_.startswith("mlm.")}
        eng = BertEncoderEngine(shape, Wp, max_tokens=n * S, max_seqs=n, max_seq_len=S, max_ranges=n, micro_batch_tokens=65536)
        eng.set_mlm_head_ex(W["mlm.dense.w"], W["mlm.dense.b"], W["mlm.ln.w"], W["mlm.ln.b"], W.get("mlm.dec.b"), W.get("mlm.dec.w"),
                            split_operands=False)
        eng.load_batch(seqs)
        tp = timed(splade)
        eng.close()
        out["splade_head"] = "split operands (default): dense as 3 accumulating GEMMs, decoder as one GEMM over K = 3H"
        out["splade_plain_operands_texts_per_s"] = n / tp
        out["splade_plain_operands_ms"] = tp * 1e3
        print(json.dumps(out))


if __name__ == "__main__":
    main()
