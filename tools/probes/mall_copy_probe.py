#!/usr/bin/env python3
"""Is the Infinity Cache visible to plain device kernels?  torch copy_/fill_/sum over buffers of growing size, looped."""
import time
import torch

dev = torch.device("cuda:0")
for mb in (8, 16, 32, 64, 96, 128, 192, 256, 512, 1024, 2048):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, device=dev, dtype=torch.float32).normal_()
    b = torch.empty_like(a)
    for name, fn, bytes_ in (("copy", lambda: b.copy_(a), 2 * mb), ("read", lambda: a.sum(), mb), ("fill", lambda: b.fill_(1.0), mb),
                             ("rmw", lambda: b.add_(1.0), 2 * mb)):
        iters = max(20, int(4000 / mb))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        print(f"{mb:5d} MB/buffer {name:5s}: {dt * 1e6:8.1f} us  {bytes_ * (1 << 20) / dt / 1e12:6.2f} TB/s", flush=True)
