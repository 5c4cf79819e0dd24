#!/usr/bin/env python3
"""Soak: repeated engine / index creation + destruction and mixed calls; reports device-memory drift.
A leak or a use-after-free in the handle lifetime code shows up here, not in the unit tests."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import verbatim_rag_amd  # noqa: E402,F401
from verbatim_rag_amd.engine import BertEncoderEngine, BertShape, EncoderEngine, ModernBertShape  # noqa: E402
from verbatim_rag_amd.vector_stores import DenseShard, SparseShard  # noqa: E402
from verbatim_rag_amd.weights import random_init, random_init_bert, random_qa_head  # noqa: E402


def used_mb():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20


def main():
    rng = np.random.default_rng(0)
    shape = ModernBertShape(vocab_size=2048, hidden_size=256, num_hidden_layers=4, num_attention_heads=4, intermediate_size=384,
                            pad_token_id=0, cls_token_id=1, sep_token_id=2)
    w = random_init(shape, 1)
    qa = random_qa_head(shape)
    bshape = BertShape(vocab_size=2048, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                       max_position_embeddings=256, cls_token_id=1, sep_token_id=2)
    bw = random_init_bert(bshape, 2)
    base = None
    t0 = time.time()
    for it in range(40):
        eng = EncoderEngine(shape, w, max_tokens=8192, max_seqs=32, max_seq_len=512, max_ranges=256,
                            micro_batch_tokens=int(rng.choice([0, 1024, 4096])))
        eng.set_qa_head(*qa)
        for _ in range(5):
            n = int(rng.integers(1, 20))
            seqs = [rng.integers(3, 2048, size=int(rng.integers(1, 400))).astype(np.int32) for _ in range(n)]
            eng.qa_logits(seqs, [[(0, len(s) - 1)] for s in seqs])
        eng.close()
        b = BertEncoderEngine(bshape, bw, max_tokens=4096, max_seqs=16, max_seq_len=256, max_ranges=16)
        seqs = [rng.integers(3, 2048, size=int(rng.integers(1, 256))).astype(np.int32) for _ in range(8)]
        b.load_batch(seqs)
        b.run()
        b.run_splade()
        b.read_splade_sparse(0.0, 2048)
        b.close()
        d = DenseShard(128, 20000, "bf16")
        d.add((rng.integers(-64, 65, size=(20000, 128)) / 64).astype(np.float32))
        d.search((rng.integers(-64, 65, size=(int(rng.integers(1, 40)), 128)) / 64).astype(np.float32), 5)
        d.close()
        nnz = rng.integers(1, 40, size=5000)
        ip = np.zeros(5001, np.int64)
        np.cumsum(nnz, out=ip[1:])
        s = SparseShard(2048, ip, rng.integers(0, 2048, size=int(ip[-1])).astype(np.int32), rng.random(int(ip[-1])).astype(np.float32))
        s.search([{int(t): 1.0 for t in rng.integers(0, 2048, 8)} for _ in range(int(rng.integers(1, 20)))], 5)
        s.close()
        torch.cuda.synchronize()
        if it == 4:
            base = used_mb()
    print(f"iterations 40, {time.time() - t0:.1f} s, device memory after warm-up {base:.0f} MiB, at the end {used_mb():.0f} MiB")


if __name__ == "__main__":
    main()
