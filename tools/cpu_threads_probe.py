#!/usr/bin/env python3
"""Which torch thread count gives the best CPU baseline for bench.py's cpu_baseline leg (B=1, 512 tokens)?"""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import torch  # noqa: E402

import verbatim_rag_amd  # noqa: E402,F401
from verbatim_rag_amd.engine import ModernBertShape  # noqa: E402
from verbatim_rag_amd.weights import random_init  # noqa: E402

shape = ModernBertShape.base()
model = bench._hf_cpu_model(shape, random_init(shape, 1234))
ids = torch.from_numpy(np.random.default_rng(0).integers(1000, 50000, size=(1, 512)))
for t in (4, 8, 16, 32, 64, 128):
    if t > (os.cpu_count() or 1):
        break
    torch.set_num_threads(t)
    with torch.no_grad():
        model(input_ids=ids)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 2.0:
            model(input_ids=ids)
            n += 1
    print(t, "threads:", round(n / (time.perf_counter() - t0), 2), "chunks/s")
