"""bf16 rows in TOPIC ORDER: a query's few thousand relevant rows sit in one run beyond the first stage, its candidate buffer overflows and the rescue
pass re-answers it.  Time of a 2-query / 64-query search with one such query in it, against the same batch without it."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import verbatim_rag_amd
from verbatim_rag_amd.vector_stores import DenseShard
n, dim, k = 1_250_000, 768, 10
rng = np.random.default_rng(0)
sh = DenseShard(dim, n, os.environ.get("ROWS", "bf16"))
topic = (rng.integers(-64, 65, size=dim) / 64.0).astype(np.float32)
for b in range(n // 125_000):
    x = (rng.integers(-64, 65, size=(125_000, dim)) / 64.0).astype(np.float32)
    if b == 5: x[20_000:26_000] += topic * 2          # 6 000 rows about the topic, rows 645 000 .. 651 000
    sh.add(x)
def timed(q, reps=10):
    sh.search(q, k); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): sh.search(q, k)
    return round((time.perf_counter() - t0) / reps * 1e3, 3)
out = {}
for nq in (2, 64):
    q = (rng.integers(-64, 65, size=(nq, dim)) / 64.0).astype(np.float32)
    out[f"plain_{nq}"] = timed(q)
    q[0] = topic
    out[f"one_topic_query_in_{nq}"] = timed(q)
    s, i = sh.search(q, k)
    out[f"topic_hits_in_run_{nq}"] = bool(((i[0] >= 645_000) & (i[0] < 651_000)).all())
print(json.dumps(out))
sh.close()
