import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import verbatim_rag_amd  # noqa: F401  (registers the hyphenated package directory)
from verbatim_rag_amd import _lib
lib = _lib.load()
dbg = _lib.load_debug()   # harness library (include/vrag_amd_debug.h)
for (M, N, K) in [(16384, 4096, 4096), (131072, 2304, 768), (131072, 2304, 1536), (32768, 2304, 768)]:
    for epi in (7, 1):
        ms = C.c_float()
        _lib.check_debug("g", dbg.vrag_debug_gemm_ms(epi, M, N, K, 20, 0, C.byref(ms)))
        print(f"epi {epi} M={M} N={N} K={K}: {ms.value*1e3:8.1f} us {2.0*M*N*K/ms.value/1e9:8.1f} TF")
