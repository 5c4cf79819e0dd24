"""Reference point only (not used by the product): vendor GEMM (torch.mm -> hipBLASLt) on the shapes of tools/gemm_bench.py,
same operand distribution (uniform [-1, 1) bf16), bf16 output.  Run it under `rocprofv3 --kernel-trace --stats` to see
which macro-tile / wave layout the vendor library picks (the kernel names spell it out)."""
import sys
import torch

shapes = [(65536, 2304, 768), (65536, 768, 768), (65536, 768, 1152), (4096, 4096, 4096), (8192, 8192, 8192)]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for M, N, K in shapes:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        torch.mm(a, w.t(), out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        torch.mm(a, w.t(), out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"vendor bf16-out  M={M} N={N} K={K} x{iters}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
