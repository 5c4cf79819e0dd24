#!/usr/bin/env python3
"""Generates tests/golden/bert_tiny.npz and distilbert_tiny.npz: HF state dicts of random-init
`BertForMaskedLM` / `DistilBertForMaskedLM` (transformers, fp32, eager attention) plus their hidden
states and MLM logits on two id sequences.  Run in the build container (needs torch + transformers);
the fixtures are data only (weights, ids, outputs).

These models are what sentence-transformers runs underneath the reference's SpladeProvider /
SentenceTransformersProvider (verbatim_rag/embedding_providers.py:52-80,117-169); sentence-transformers
itself is not installed here, so the pooling on top stays a restatement.
"""
import os

import numpy as np
import torch
from transformers import (BertConfig, BertForMaskedLM, BertForSequenceClassification, DistilBertConfig,
                          DistilBertForMaskedLM)

HERE = os.path.dirname(os.path.abspath(__file__))


def perturb(model, seed):
    """Default init zeroes biases and sets LayerNorm to (1, 0): randomise them so bias paths are pinned."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            elif "LayerNorm.weight" in name or "layer_norm.weight" in name:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() == 2:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    return model


def run(model, base, ids_list):
    outs = {}
    model.eval()
    with torch.no_grad():
        for i, ids in enumerate(ids_list):
            t = torch.tensor([ids], dtype=torch.long)
            hid = base(input_ids=t).last_hidden_state[0].numpy()
            logits = model(input_ids=t).logits[0].numpy()
            outs[f"ids{i}"] = np.asarray(ids, dtype=np.int32)
            outs[f"hidden{i}"] = hid.astype(np.float32)
            outs[f"mlm{i}"] = logits.astype(np.float32)
    return outs


def main():
    rng = np.random.default_rng(7)
    V, H, L, NH, I, P = 512, 128, 2, 2, 256, 64
    ids_list = [rng.integers(5, V, size=37).tolist(), rng.integers(5, V, size=64).tolist()]

    torch.manual_seed(11)
    cfg = BertConfig(vocab_size=V, hidden_size=H, num_hidden_layers=L, num_attention_heads=NH, intermediate_size=I,
                     max_position_embeddings=P, type_vocab_size=2, hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0, attn_implementation="eager")
    m = perturb(BertForMaskedLM(cfg), 1)
    m.tie_weights()
    out = run(m, m.bert, ids_list)
    sd = {"sd:" + k: v.detach().numpy().astype(np.float32) for k, v in m.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "bert_tiny.npz"), cfg=np.asarray([V, H, L, NH, I, P], dtype=np.int32), **sd, **out)

    torch.manual_seed(12)
    dcfg = DistilBertConfig(vocab_size=V, dim=H, n_layers=L, n_heads=NH, hidden_dim=I, max_position_embeddings=P,
                            dropout=0.0, attention_dropout=0.0, attn_implementation="eager")
    d = perturb(DistilBertForMaskedLM(dcfg), 2)
    d.tie_weights()
    out = run(d, d.distilbert, ids_list)
    sd = {"sd:" + k: v.detach().numpy().astype(np.float32) for k, v in d.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "distilbert_tiny.npz"), cfg=np.asarray([V, H, L, NH, I, P], dtype=np.int32), **sd, **out)
    # cross-encoder (BertForSequenceClassification, num_labels 1, MiniLM-like head_dim 32, sentence pairs with token types)
    torch.manual_seed(13)
    ccfg = BertConfig(vocab_size=V, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256,
                      max_position_embeddings=P, type_vocab_size=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                      num_labels=1, attn_implementation="eager")
    c = perturb(BertForSequenceClassification(ccfg), 3)
    c.eval()
    out = {}
    pairs = [(9, 28), (20, 40), (3, 5)]
    with torch.no_grad():
        for i, (nq, nd) in enumerate(pairs):
            ids = [1] + rng.integers(5, V, size=nq).tolist() + [2] + rng.integers(5, V, size=nd).tolist() + [2]
            tt = [0] * (nq + 2) + [1] * (nd + 1)
            r = c(input_ids=torch.tensor([ids]), token_type_ids=torch.tensor([tt]), output_hidden_states=True)
            out[f"ids{i}"], out[f"types{i}"] = np.asarray(ids, np.int32), np.asarray(tt, np.int32)
            out[f"hidden{i}"] = r.hidden_states[-1][0].numpy().astype(np.float32)
            out[f"logits{i}"] = r.logits[0].numpy().astype(np.float32)
    sd = {"sd:" + k: v.detach().numpy().astype(np.float32) for k, v in c.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "bert_pair_tiny.npz"), cfg=np.asarray([V, 128, 2, 4, 256, P], dtype=np.int32), **sd, **out)
    print("wrote bert_tiny.npz, distilbert_tiny.npz, bert_pair_tiny.npz")


if __name__ == "__main__":
    main()
