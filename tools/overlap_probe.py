"""Hardware probe: do an MFMA-bound kernel and an HBM-bound kernel overlap when issued on two streams?"""
import time
import torch
a = torch.randn(16384, 4096, device="cuda", dtype=torch.bfloat16)
b = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
x = torch.zeros(131072 * 768, device="cuda", dtype=torch.float32)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def mm(n):
    with torch.cuda.stream(s1):
        for _ in range(n):
            torch.mm(a, b)
def rmw(n):
    with torch.cuda.stream(s2):
        for _ in range(n):
            x.add_(1.0)
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
mm(5); rmw(5)
tm = t(lambda: mm(40)); tr = t(lambda: rmw(40))
tb = t(lambda: (mm(40), rmw(40)))
fl = 2 * 16384 * 4096 * 4096 * 40
print(f"mm x40: {tm:.2f} ms ({fl/tm/1e9:.0f} TF/s)  rmw x40: {tr:.2f} ms ({806e6*40/tr/1e9:.2f} TB/s)  both: {tb:.2f} ms  (sum {tm+tr:.2f}, max {max(tm,tr):.2f})")
