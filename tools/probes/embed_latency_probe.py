"""Single-question latency of the SPLADE query provider (BERT-base shape, V = 30 522, fp16 operands, split head): phase split per call."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tokenizers import Tokenizer
import verbatim_rag_amd
from verbatim_rag_amd.embedding_providers import GpuSpladeProvider
from verbatim_rag_amd.engine import BertEncoderEngine, BertShape
from verbatim_rag_amd.weights import random_init_bert
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tok = Tokenizer.from_file(os.path.join(ROOT, "tests", "golden", "tokenizer.json"))
V = 30522
bshape = BertShape.bert_base()
W = random_init_bert(bshape, seed=1234)
emb = BertEncoderEngine(bshape, {k: v for k, v in W.items() if not k.startswith("mlm.")}, max_tokens=32768, max_seqs=2048, max_seq_len=128,
                        max_ranges=2048, device=0, operand_dtype="f16")
head = (W["mlm.dense.w"], W["mlm.dense.b"], W["mlm.ln.w"], W["mlm.ln.b"])
emb.set_mlm_head_ex(*head, np.full(V, -3.0, np.float32), W.get("mlm.dec.w"))
prov = GpuSpladeProvider(emb, tok, max_length=128, sparse_cap=4096)
qs = ["Where is the tall iron tower number %d in the old city?" % i for i in range(300)]
for q in qs[:20]: prov.embed_queries([q])
t0 = time.perf_counter()
for q in qs[20:]: prov.embed_queries([q])
total = (time.perf_counter() - t0) / 280
ph = {"encode": 0.0, "load_batch": 0.0, "run": 0.0, "run_splade": 0.0, "read_sparse": 0.0, "clamp_check": 0.0}
for q in qs[20:]:
    a = time.perf_counter(); seqs = prov._encode([q]); b = time.perf_counter(); ph["encode"] += b - a
    emb.load_batch(seqs); c = time.perf_counter(); ph["load_batch"] += c - b
    emb.run(); torch.cuda.synchronize(); d = time.perf_counter(); ph["run"] += d - c
    emb.run_splade(); torch.cuda.synchronize(); e = time.perf_counter(); ph["run_splade"] += e - d
    emb.read_splade_sparse(1e-6, 4096); f = time.perf_counter(); ph["read_sparse"] += f - e
    prov._f16_clamped(); g = time.perf_counter(); ph["clamp_check"] += g - f
print(json.dumps({"embed_queries_one_ms": round(total * 1e3, 3), "tokens": len(seqs[0]), "phases_ms_synced": {k: round(v / 280 * 1e3, 3) for k, v in ph.items()}}))
emb.close()
