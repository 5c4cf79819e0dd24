"""[Historical probe: the VRAG_TOPK_STAGE_RATIO knob it drove was removed in round 6 once its sweep was recorded in
profiles/r05_tiled_stage_ratio_probe.txt; re-add `ratio_env` in csrc/topk.hip dense_tiled_search to repeat it.]
Stage growth of the tiled dense search (VRAG_TOPK_STAGE_RATIO) against the time of a batched search: fp32 rows with the prefilter
image (64 candidates per query from the image) and bf16 rows (k = 10), 1.25 M x 768, through the public call.
  for r in 4 8 16; do VRAG_TOPK_STAGE_RATIO=$r python tools/probes/tiled_stage_ratio_probe.py; done"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import verbatim_rag_amd  # noqa: F401,E402
from verbatim_rag_amd.vector_stores import DenseShard  # noqa: E402

n, dim, k = 1_250_000, 768, 10
out = {"stage_ratio": os.environ.get("VRAG_TOPK_STAGE_RATIO", "default")}
for dtype in ("f32", "bf16"):
    sh = DenseShard(dim, n, dtype)
    g = torch.Generator(device="cuda").manual_seed(1)
    for a in range(0, n, 125_000):
        slab = torch.randn((125_000, dim), generator=g, device="cuda", dtype=torch.float32)
        sh.add_device(slab.data_ptr(), 125_000, None)
        del slab
    for nq in (64, 256, 1024):
        q = np.random.default_rng(nq).standard_normal((nq, dim)).astype(np.float32)
        sh.search(q, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            sh.search(q, k)
        out[f"{dtype}_{nq}q_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    sh.close()
print(json.dumps(out), flush=True)
