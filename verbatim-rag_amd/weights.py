"""Weight sources for the engine: HF safetensors directories (the reference's checkpoint
format, extractor_models/model.py:126-151 / trainer.py:468-494) and seeded random init
(no checkpoint can be downloaded in the build/bench environment)."""
from __future__ import annotations

import json
import math
import os
from typing import Dict, Optional, Tuple

import numpy as np

from .engine import ModernBertShape


def _trunc_normal(rng: np.random.Generator, shape, std: float, cutoff: float = 3.0) -> np.ndarray:
    x = rng.standard_normal(size=shape, dtype=np.float32)
    bad = np.abs(x) > cutoff
    while bad.any():
        x[bad] = rng.standard_normal(size=int(bad.sum()), dtype=np.float32)
        bad = np.abs(x) > cutoff
    x *= np.float32(std)
    return x


def random_init(shape: ModernBertShape, seed: int = 1234, init_range: float = 0.02) -> Dict[str, np.ndarray]:
    """Trunc-normal init with the reference architecture's std scheme (transformers
    modeling_modernbert.py:352-400: 'in' 0.02, 'out' 0.02/sqrt(2L)); LayerNorm gains are
    jittered around 1 so a kernel that drops them is caught by parity tests."""
    rng = np.random.default_rng(seed)
    H, I, L = shape.hidden_size, shape.intermediate_size, shape.num_hidden_layers
    out_std = init_range / math.sqrt(2.0 * L)
    w: Dict[str, np.ndarray] = {}
    w["embeddings.tok_embeddings.weight"] = _trunc_normal(rng, (shape.vocab_size, H), init_range)
    w["embeddings.norm.weight"] = (1.0 + 0.1 * rng.standard_normal(H)).astype(np.float32)
    for l in range(L):
        p = f"layers.{l}."
        if l > 0:
            w[p + "attn_norm.weight"] = (1.0 + 0.1 * rng.standard_normal(H)).astype(np.float32)
        w[p + "attn.Wqkv.weight"] = _trunc_normal(rng, (3 * H, H), init_range)
        w[p + "attn.Wo.weight"] = _trunc_normal(rng, (H, H), out_std)
        w[p + "mlp_norm.weight"] = (1.0 + 0.1 * rng.standard_normal(H)).astype(np.float32)
        w[p + "mlp.Wi.weight"] = _trunc_normal(rng, (2 * I, H), init_range)
        w[p + "mlp.Wo.weight"] = _trunc_normal(rng, (H, I), out_std)
    w["final_norm.weight"] = (1.0 + 0.1 * rng.standard_normal(H)).astype(np.float32)
    return w


def random_qa_head(shape: ModernBertShape, seed: int = 4321, num_labels: int = 2) -> Tuple[np.ndarray, np.ndarray]:
    rng = np.random.default_rng(seed)
    w = _trunc_normal(rng, (num_labels, shape.hidden_size), shape.hidden_size ** -0.5)
    b = (0.1 * rng.standard_normal(num_labels)).astype(np.float32)
    return w, b


def load_safetensors_dir(path: str) -> Tuple[ModernBertShape, Dict[str, np.ndarray], dict]:
    """Reads config.json + *.safetensors from a local HF checkpoint directory."""
    from safetensors.numpy import load_file

    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    tensors: Dict[str, np.ndarray] = {}
    for fn in sorted(os.listdir(path)):
        if fn.endswith(".safetensors"):
            tensors.update(load_file(os.path.join(path, fn)))
    if not tensors:
        raise FileNotFoundError(f"no .safetensors file under {path}")
    return ModernBertShape.from_hf_config(cfg), {k: np.asarray(v, dtype=np.float32) for k, v in tensors.items()}, cfg
