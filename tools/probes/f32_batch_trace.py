"""Kernel-level breakdown of ONE batched search over fp32 rows with the prefilter image (run under rocprofv3 --kernel-trace):
  cd /tmp && rocprofv3 --kernel-trace --stats -d out -- python tools/probes/f32_batch_trace.py [nq]
The shard is built before the marker kernel-free section; only `reps` searches of `nq` queries follow."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import verbatim_rag_amd  # noqa: F401,E402
from verbatim_rag_amd.vector_stores import DenseShard  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n, dim, k, reps = 1_250_000, 768, 10, 20
sh = DenseShard(dim, n, "f32")
g = torch.Generator(device="cuda").manual_seed(1)
for a in range(0, n, 125_000):
    slab = torch.randn((125_000, dim), generator=g, device="cuda", dtype=torch.float32)
    sh.add_device(slab.data_ptr(), 125_000, None)
q = np.random.default_rng(0).standard_normal((nq, dim)).astype(np.float32)
for _ in range(reps):
    sh.search(q, k)
torch.cuda.synchronize()
sh.close()
