import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import verbatim_rag_amd
from verbatim_rag_amd import _lib
lib = _lib.load()
dbg = _lib.load_debug()   # harness library (include/vrag_amd_debug.h)
for epi, name, N, K in ((3, "resid", 768, 768), (3, "resid", 768, 1152), (5, "qkv", 2304, 768), (4, "geglu", 2304, 768)):
    for M in (1024, 2048):
        ms = C.c_float()
        _lib.check_debug("gemm", dbg.vrag_debug_gemm_ms(epi, M, N, K, 2000, 0, C.byref(ms)))
        print(f"{name:6s} M={M:5d} N={N:5d} K={K:5d}: {ms.value*1e3:7.2f} us", flush=True)
