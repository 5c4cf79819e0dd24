"""Weight sources for the engine: HF safetensors directories (the reference's checkpoint
format, extractor_models/model.py:126-151 / trainer.py:468-494) and seeded random init
(no checkpoint can be downloaded in the build/bench environment)."""
from __future__ import annotations

import json
import math
import os
from typing import Dict, Tuple

import numpy as np

from .engine import BertShape, ModernBertShape


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


# ----------------------------------------------------------------------------- BERT family
def bert_canonical(tensors: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """HF BERT / DistilBERT tensor names (BertModel, BertForMaskedLM, DistilBertModel, DistilBertForMaskedLM; with
    or without the `bert.` / `distilbert.` prefix; transformers models/bert/modeling_bert.py,
    models/distilbert/modeling_distilbert.py) -> the flat names `BertEncoderEngine` takes.  Query / key / value
    matrices and biases are concatenated along the output dimension; `emb.type0` is row 0 of the token-type
    table (single-segment inputs, which is all the embedding providers send)."""
    t: Dict[str, np.ndarray] = {}
    for k, v in tensors.items():
        for p in ("bert.", "distilbert."):
            if k.startswith(p):
                k = k[len(p):]
                break
        t[k] = np.asarray(v, dtype=np.float32)
    W: Dict[str, np.ndarray] = {}
    distil = any(k.startswith("transformer.layer.") for k in t)
    lp = "transformer.layer." if distil else "encoder.layer."
    layer_ids = sorted({int(k[len(lp):].split(".")[0]) for k in t if k.startswith(lp)})
    if not layer_ids:
        raise KeyError("no BERT / DistilBERT layer tensors found")
    W["emb.word"] = t["embeddings.word_embeddings.weight"]
    W["emb.pos"] = t["embeddings.position_embeddings.weight"]
    W["emb.ln.w"], W["emb.ln.b"] = t["embeddings.LayerNorm.weight"], t["embeddings.LayerNorm.bias"]
    if not distil:
        W["emb.type0"] = np.ascontiguousarray(t["embeddings.token_type_embeddings.weight"][0])
        W["emb.types"] = t["embeddings.token_type_embeddings.weight"]       # sentence pairs (cross-encoders)
        if "pooler.dense.weight" in t:
            W["pooler.w"], W["pooler.b"] = t["pooler.dense.weight"], t["pooler.dense.bias"]
        if "classifier.weight" in t:                                        # BertForSequenceClassification
            W["cls.w"], W["cls.b"] = t["classifier.weight"], t["classifier.bias"]
    names = ({"q": "attention.q_lin", "k": "attention.k_lin", "v": "attention.v_lin", "o": "attention.out_lin",
              "ln1": "sa_layer_norm", "w1": "ffn.lin1", "w2": "ffn.lin2", "ln2": "output_layer_norm"} if distil else
             {"q": "attention.self.query", "k": "attention.self.key", "v": "attention.self.value",
              "o": "attention.output.dense", "ln1": "attention.output.LayerNorm", "w1": "intermediate.dense",
              "w2": "output.dense", "ln2": "output.LayerNorm"})
    for i in layer_ids:
        p = f"{lp}{i}."
        W[f"l{i}.wqkv"] = np.concatenate([t[p + names[n] + ".weight"] for n in "qkv"], axis=0)
        W[f"l{i}.bqkv"] = np.concatenate([t[p + names[n] + ".bias"] for n in "qkv"], axis=0)
        W[f"l{i}.wo"], W[f"l{i}.bo"] = t[p + names["o"] + ".weight"], t[p + names["o"] + ".bias"]
        W[f"l{i}.ln1.w"], W[f"l{i}.ln1.b"] = t[p + names["ln1"] + ".weight"], t[p + names["ln1"] + ".bias"]
        W[f"l{i}.w1"], W[f"l{i}.b1"] = t[p + names["w1"] + ".weight"], t[p + names["w1"] + ".bias"]
        W[f"l{i}.w2"], W[f"l{i}.b2"] = t[p + names["w2"] + ".weight"], t[p + names["w2"] + ".bias"]
        W[f"l{i}.ln2.w"], W[f"l{i}.ln2.b"] = t[p + names["ln2"] + ".weight"], t[p + names["ln2"] + ".bias"]
    mlm = (("vocab_transform", "vocab_layer_norm", "vocab_projector.bias") if distil else
           ("cls.predictions.transform.dense", "cls.predictions.transform.LayerNorm", "cls.predictions.bias"))
    if mlm[0] + ".weight" in t:   # decoder weight is tied to the word embeddings in both families
        W["mlm.dense.w"], W["mlm.dense.b"] = t[mlm[0] + ".weight"], t[mlm[0] + ".bias"]
        W["mlm.ln.w"], W["mlm.ln.b"] = t[mlm[1] + ".weight"], t[mlm[1] + ".bias"]
        W["mlm.dec.b"] = t[mlm[2]] if mlm[2] in t else t.get("cls.predictions.decoder.bias")
    return W


def random_init_bert(shape: BertShape, seed: int = 1234, std: float = 0.02, mlm: bool = True) -> Dict[str, np.ndarray]:
    """Seeded random weights in the flat naming (no checkpoint can be downloaded in the build / bench environment):
    normal(0, std) matrices (transformers models/bert/modeling_bert.py:541-553 initializer_range), LayerNorm gains
    jittered around 1 and small non-zero biases so a kernel that drops one is caught by parity tests."""
    rng = np.random.default_rng(seed)
    H, I, V, P = shape.hidden_size, shape.intermediate_size, shape.vocab_size, shape.max_position_embeddings

    def n(*shp, s=std):
        return (rng.standard_normal(size=shp, dtype=np.float32) * np.float32(s))

    def gain():
        return (1.0 + 0.1 * rng.standard_normal(H)).astype(np.float32)

    W = {"emb.word": n(V, H), "emb.pos": n(P, H), "emb.ln.w": gain(), "emb.ln.b": n(H)}
    if shape.model_type == "bert":
        W["emb.type0"] = n(H)
    for i in range(shape.num_hidden_layers):
        W[f"l{i}.wqkv"], W[f"l{i}.bqkv"] = n(3 * H, H), n(3 * H)
        W[f"l{i}.wo"], W[f"l{i}.bo"] = n(H, H), n(H)
        W[f"l{i}.ln1.w"], W[f"l{i}.ln1.b"] = gain(), n(H)
        W[f"l{i}.w1"], W[f"l{i}.b1"] = n(I, H), n(I)
        W[f"l{i}.w2"], W[f"l{i}.b2"] = n(H, I), n(H)
        W[f"l{i}.ln2.w"], W[f"l{i}.ln2.b"] = gain(), n(H)
    if mlm:
        W["mlm.dense.w"], W["mlm.dense.b"] = n(H, H), n(H)
        W["mlm.ln.w"], W["mlm.ln.b"] = gain(), n(H)
        W["mlm.dec.b"] = n(V)
    return W


def load_bert_safetensors_dir(path: str) -> Tuple[BertShape, Dict[str, np.ndarray], dict]:
    """config.json + *.safetensors of a local BERT / DistilBERT checkpoint (e.g. a downloaded `naver/splade-v3`
    or `BAAI/bge-base-en-v1.5` snapshot) -> (shape, flat weights, raw config)."""
    from safetensors.numpy import load_file

    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    if cfg.get("model_type") not in ("bert", "distilbert"):
        raise ValueError(f"{path}: model_type {cfg.get('model_type')!r} is not bert / distilbert")
    tensors: Dict[str, np.ndarray] = {}
    for fn in sorted(os.listdir(path)):
        if fn.endswith(".safetensors"):
            tensors.update(load_file(os.path.join(path, fn)))
    if not tensors:
        raise FileNotFoundError(f"no .safetensors file under {path}")
    return BertShape.from_hf_config(cfg), bert_canonical(tensors), cfg
