"""Reference point only (not used by the product): vendor GEMM (torch.mm -> hipBLASLt) on the ModernBERT shapes."""
import time
import torch
M = 131072
for N, K in [(2304, 768), (768, 768), (768, 1152)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16) * 0.5
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.5
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        torch.mm(a, w.t(), out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        torch.mm(a, w.t(), out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"vendor bf16-out  M={M} N={N} K={K}: {dt*1e6:8.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TFLOP/s")
