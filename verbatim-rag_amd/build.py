"""Builds libvrag_amd.so (gfx950) in-tree with hipcc. No JIT cache, no torch extension:
the library is a plain C-ABI shared object (include/vrag_amd.h)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# VRAG_BUILD_VARIANT=<tag> (tuning experiments only): objects under build_<tag>/, library libvrag_amd_<tag>.so -- load it with VRAG_AMD_LIB
VARIANT = os.environ.get("VRAG_BUILD_VARIANT", "")
LIB_PATH = os.path.join(HERE, f"libvrag_amd_{VARIANT}.so" if VARIANT else "libvrag_amd.so")
SOURCES = ["gemm_bf16.hip", "attention.hip", "qkv_attn.hip", "norm_heads.hip", "topk.hip", "text.hip", "comm.hip", "capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
FLAGS += os.environ.get("VRAG_HIPCC_FLAGS", "").split()  # tuning experiments only


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libvrag_amd.so)")


def _newer(src: str, dst: str) -> bool:
    return not os.path.exists(dst) or os.path.getmtime(src) > os.path.getmtime(dst)


DEBUG_LIB_PATH = os.path.join(HERE, f"libvrag_amd_dbg_{VARIANT}.so" if VARIANT else "libvrag_amd_dbg.so")
# Harness build (include/vrag_amd_debug.h): the product objects, except that the fused kernel keeps its phase-decomposition
# branches (-DVRAG_DEBUG_API), plus the synthetic-operand timing loops / unit-test hook of debug_api.hip.
DEBUG_ONLY = ["debug_api.hip"]
DEBUG_RECOMPILED = ["qkv_attn.hip"]


def build_library(force: bool = False, verbose: bool = False, debug: bool = True) -> str:
    """Builds libvrag_amd.so (the product) and, with debug=True, libvrag_amd_dbg.so (the tuning / unit-test harness)."""
    hipcc = _hipcc()
    objdir = os.path.join(HERE, f"build_{VARIANT}" if VARIANT else "build")
    os.makedirs(objdir, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(os.path.dirname(HERE), "include", h) for h in ("vrag_amd.h", "vrag_amd_debug.h")]
    hdr_time = max(os.path.getmtime(h) for h in headers)

    def compile_one(job) -> str:
        src, suffix, extra = job
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".hip", suffix + ".o"))
        if force or _newer(sp, obj) or os.path.getmtime(obj) < hdr_time:
            cmd = [hipcc, *FLAGS, *extra, "-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    jobs = [(s, "", []) for s in srcs]
    if debug:
        jobs += [(s, "_dbg", ["-DVRAG_DEBUG_API"]) for s in DEBUG_RECOMPILED + DEBUG_ONLY]
    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        objs = list(ex.map(compile_one, jobs))
    prod = objs[: len(srcs)]

    def link(inputs, out):
        if force or any(_newer(o, out) for o in inputs):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *inputs, "-ldl", "-o", out]   # -ldl: comm.hip binds RCCL at run time
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")

    link(prod, LIB_PATH)
    if debug:
        swapped = {os.path.join(objdir, s.replace(".hip", ".o")): os.path.join(objdir, s.replace(".hip", "_dbg.o")) for s in DEBUG_RECOMPILED}
        dbg = [swapped.get(o, o) for o in prod] + [os.path.join(objdir, s.replace(".hip", "_dbg.o")) for s in DEBUG_ONLY]
        link(dbg, DEBUG_LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
