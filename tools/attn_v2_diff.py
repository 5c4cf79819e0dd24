#!/usr/bin/env python3
"""Localises where the second-generation attention kernel (VRAG_ATTN_V2=1) departs from the first: runs 1 and 2 encoder layers
(layer 0 global, layer 1 banded) of the tiny test geometry on several sequence lengths in two subprocesses and prints, per
case, the largest hidden-state difference and the rows where it sits, next to each kernel's distance from the fp32 oracle."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = [(64,), (200,), (512,), (130, 77), (1000,)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np

    import verbatim_rag_amd  # noqa: F401
    from oracle import modernbert_np as O
    from verbatim_rag_amd.engine import EncoderEngine, ModernBertShape

    tiny = dict(vocab_size=512, hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=192,
                pad_token_id=0, cls_token_id=1, sep_token_id=2)
    w = O.random_weights(O.EncoderConfig(**tiny), seed=7)
    eng = EncoderEngine(ModernBertShape(**tiny), w, max_tokens=8192, max_seqs=16, max_seq_len=2048, max_ranges=16)
    out = {}
    for ci, lens in enumerate(CASES):
        rng = np.random.default_rng(ci)
        seqs = [rng.integers(3, 512, size=n).astype(np.int32) for n in lens]
        for nl in (1, 2):
            eng.load_batch(seqs)
            eng.run(n_layers=nl)
            out[f"{ci}_{nl}"] = eng.read_hidden(final_norm=False).tolist()
    eng.close()
    json.dump(out, open(sys.argv[2], "w"))
    sys.exit(0)

import numpy as np  # noqa: E402

from oracle import modernbert_np as O  # noqa: E402

res = {}
VARIANTS = {"v2": {"VRAG_ATTN_V2": "1"}, "v2 lazy=0": {"VRAG_ATTN_V2": "1", "VRAG_ATTN_V2_LAZY": "0"},
            "v2 no seed": {"VRAG_ATTN_V2": "1", "VRAG_ATTN_V2_NOSEED": "1"},
            "v2 lazy=0 no seed": {"VRAG_ATTN_V2": "1", "VRAG_ATTN_V2_LAZY": "0", "VRAG_ATTN_V2_NOSEED": "1"}}
for tag, env in (("v1", {}), *VARIANTS.items()):
    path = f"/tmp/attn_{tag.replace(' ', '_').replace('=', '')}.json"
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", path], check=True, env={**os.environ, **env})
    res[tag] = {k: np.asarray(v, np.float32) for k, v in json.load(open(path)).items()}
tiny = dict(vocab_size=512, hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=192,
            pad_token_id=0, cls_token_id=1, sep_token_id=2)
ocfg = O.EncoderConfig(**tiny)
ow = O.random_weights(ocfg, seed=7)
oracle = {}
for ci, lens in enumerate(CASES):
    rng = np.random.default_rng(ci)
    seqs = [rng.integers(3, 512, size=n).astype(np.int32) for n in lens]
    per = [O.encoder_forward(ocfg, ow, s_, return_all=True)[1] for s_ in seqs]
    for nl in (1, 2):
        oracle[f"{ci}_{nl}"] = np.concatenate([hs[nl] for hs in per])
for key in res["v1"]:
    a, b = res["v1"][key], res["v2"][key]
    ref = oracle[key]
    ea, eb = np.abs(a - ref).max(axis=1), np.abs(b - ref).max(axis=1)
    print(f"   vs oracle: v1 max {ea.max():.3e} (row {int(ea.argmax())}), v2 max {eb.max():.3e} (row {int(eb.argmax())}); "
          f"v2 rows over 3x v1's max: {np.nonzero(eb > 3 * ea.max())[0].tolist()[:24]}")
    d = np.abs(a - b).max(axis=1)
    worst = np.argsort(-d)[:8]
    print(f"case {key} lens={CASES[int(key.split('_')[0])]} layers={key.split('_')[1]}: max |v1 - v2| = {d.max():.3e} at rows {worst.tolist()} "
          f"({(d > 0.1 * d.max()).sum()} rows above a tenth of it); |hidden| ~ {np.abs(a).mean():.3f}")

for tag in VARIANTS:
    worst = {key: float(np.abs(res[tag][key] - oracle[key]).max()) for key in res[tag] if key.endswith("_2")}
    print(f"{tag:20s} two layers, max |hidden - oracle| per case: " + " ".join(f"{v:.2e}" for v in worst.values()))
print(f"{'v1':20s} two layers, max |hidden - oracle| per case: " + " ".join(f"{float(np.abs(res['v1'][k] - oracle[k]).max()):.2e}" for k in res["v1"] if k.endswith("_2")))
