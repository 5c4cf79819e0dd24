#!/usr/bin/env python3
"""Joules per launch of every kernel class of the extractor step, each looped ALONE on the chip while the package power is
sampled (the step sits on the package power limit, so this table is its roofline -- time at the cap =
dynamic energy / (cap - static power)).  Classes: the three fused GEMM epilogues and the bare main loops of the same
shapes (EPI_NONE), global and banded attention; ModernBERT-base shapes at one 65 536-token micro-batch.

  python tools/energy_by_class.py [--seconds 5] [--out gpurun_out/energy_by_class.json]
"""
import argparse
import ctypes as C
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verbatim_rag_amd  # noqa: E402,F401
from verbatim_rag_amd import _lib  # noqa: E402

POWER_RE = re.compile(r"Current Socket Graphics Package Power \(W\):\s*([0-9.]+)")
SCLK_RE = re.compile(r"sclk clock level:.*\((\d+)Mhz\)")
LAST_SCLK = [None]


def _hwmon():
    """The hwmon power file is not usable on this part (it sat at its idle value through every loop, r3a session): the
    package power comes from `rocm-smi --showpower` like tools/energy_probe.py, unless VRAG_POWER_HWMON names a file."""
    return os.environ.get("VRAG_POWER_HWMON") or None


def read_power(hw):
    if hw:
        try:
            return int(open(hw).read().strip()) / 1e6
        except Exception:
            pass
    try:
        txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        m = POWER_RE.search(txt)
        c = SCLK_RE.search(txt)
        LAST_SCLK[0] = float(c.group(1)) if c else None
        return float(m.group(1)) if m else None
    except Exception:
        return None


class Sampler(threading.Thread):
    def __init__(self, hw, interval):
        super().__init__(daemon=True)
        self.hw, self.interval, self.samples, self.stop = hw, interval, [], threading.Event()
        self.clocks = []

    def run(self):
        while not self.stop.is_set():
            w = read_power(self.hw)
            if w is not None:
                self.samples.append((time.perf_counter(), w))
                self.clocks.append((time.perf_counter(), LAST_SCLK[0]))
            self.stop.wait(self.interval)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=5.0)
    ap.add_argument("--tokens", type=int, default=65536)
    ap.add_argument("--out", default="gpurun_out/energy_by_class.json")
    ap.add_argument("--only", default="", help="comma-separated substrings of class names to run")
    args = ap.parse_args()
    lib = _lib.load()
    dbg = _lib.load_debug()   # harness library (include/vrag_amd_debug.h)
    hw = _hwmon()
    M, H, I, S = args.tokens, 768, 1152, 512
    classes = [   # name, kind, args, algorithmic FLOP per launch
        ("gemm_qkv (EPI_QKV_ROPE, N=2304 K=768)", "gemm", (5, M, 3 * H, H), 2.0 * M * 3 * H * H),
        ("gemm_wi (EPI_GEGLU, N=2304 K=768)", "gemm", (4, M, 2 * I, H), 2.0 * M * 2 * I * H),
        ("mainloop N=2304 K=768 (EPI_NONE)", "gemm", (7, M, 3 * H, H), 2.0 * M * 3 * H * H),
        ("bf16-out N=2304 K=768 (EPI_BF16)", "gemm", (1, M, 3 * H, H), 2.0 * M * 3 * H * H),
        ("gemm_wo (EPI_RESIDUAL, N=768 K=768)", "gemm", (3, M, H, H), 2.0 * M * H * H),
        ("mainloop N=768 K=768 (EPI_NONE)", "gemm", (7, M, H, H), 2.0 * M * H * H),
        ("gemm_wo_mlp (EPI_RESIDUAL, N=768 K=1152)", "gemm", (3, M, H, I), 2.0 * M * H * I),
        ("mainloop N=768 K=1152 (EPI_NONE)", "gemm", (7, M, H, I), 2.0 * M * H * I),
        # the residual epilogue (almost) alone: one K-step of main loop in front of it -- what does the read-modify-write phase draw?
        ("resid epilogue-only (EPI_RESIDUAL, N=768 K=64)", "gemm", (3, M, H, 64), 2.0 * M * H * 64),
        ("resid plain epilogue-only (no fold outputs, N=768 K=64)", "gemm_plain", (3, M, H, 64), 2.0 * M * H * 64),
        ("resid split epilogue-only (operand plane + remainder byte, N=768 K=64)", "gemm_split", (3, M, H, 64), 2.0 * M * H * 64),
        ("gemm_wo split (EPI_RESIDUAL, N=768 K=768)", "gemm_split", (3, M, H, H), 2.0 * M * H * H),
        ("gemm_wo_mlp split (EPI_RESIDUAL, N=768 K=1152)", "gemm_split", (3, M, H, I), 2.0 * M * H * I),
        ("geglu epilogue-only (EPI_GEGLU, N=2304 K=64)", "gemm", (4, M, 2 * I, 64), 2.0 * M * 2 * I * 64),
        ("attn_global (S=512)", "attn", (0, M // S, S, H, 64), 4.0 * M * S * H),
        ("attn_local (S=512, |i-j|<=64)", "attn", (1, M // S, S, H, 64), 4.0 * M * 129 * H),
        # the fused Wqkv + RoPE + attention kernel (csrc/qkv_attn.hip): replaces gemm_qkv + attn_* of its layers
        ("qkv_attn fused global (S=512)", "fused", (0, M // S, S, H, 64), 2.0 * M * 3 * H * H + 4.0 * M * S * H),
        ("qkv_attn fused banded (S=512, |i-j|<=64)", "fused", (1, M // S, S, H, 64), 2.0 * M * 3 * H * H + 4.0 * M * 129 * H),
    ]
    if args.only:
        classes = [c for c in classes if any(t in c[0] for t in args.only.split(","))]
    idle = [read_power(hw) for _ in range(5) if time.sleep(0.2) is None]
    idle_w = sum(w for w in idle if w) / max(1, len([w for w in idle if w]))
    out = {"power_source": hw or "rocm-smi --showpower", "idle_w": idle_w, "tokens_per_launch": M, "classes": {}}

    def run(kind, a, iters):
        ms = C.c_float()
        if kind in ("gemm_plain", "gemm_split"):
            var = "VRAG_DEBUG_GEMM_PLAIN_RESID" if kind == "gemm_plain" else "VRAG_DEBUG_GEMM_SPLIT"
            os.environ[var] = "1"
            try:
                _lib.check_debug("gemm", dbg.vrag_debug_gemm_ms(a[0], a[1], a[2], a[3], iters, 0, C.byref(ms)))
            finally:
                del os.environ[var]
        elif kind == "gemm":
            _lib.check_debug("gemm", dbg.vrag_debug_gemm_ms(a[0], a[1], a[2], a[3], iters, 0, C.byref(ms)))
        elif kind == "fused":
            _lib.check_debug("fused", dbg.vrag_debug_qkv_attn_ms(a[0], a[1], a[2], a[3], a[4], iters, 0, 0, C.byref(ms)))
        else:
            _lib.check_debug("attn", dbg.vrag_debug_attn_ms(a[0], a[1], a[2], a[3], a[4], iters, 0, C.byref(ms)))
        return ms.value

    for name, kind, a, flop in classes:
        ms = run(kind, a, 50)                                        # calibrate the launch count
        iters = max(200, int(args.seconds / (ms * 1e-3)))
        smp = Sampler(hw, 0.05 if hw else 0.05)          # rocm-smi itself takes a few hundred ms per reading
        t0 = time.perf_counter()
        smp.start()
        ms = run(kind, a, iters)
        t1 = time.perf_counter()
        smp.stop.set()
        smp.join()
        # the loop itself occupies [t1 - iters * ms, t1]; drop the first 30 % (sensor averaging window, allocation, warm-up)
        lo = t1 - iters * ms * 1e-3 * 0.7
        busy = [w for t, w in smp.samples if lo <= t <= t1]
        avg = sum(busy) / len(busy) if busy else None
        clk = [c for t, c in smp.clocks if lo <= t <= t1 and c]
        rec = {"us_per_launch": ms * 1e3, "launches": iters, "tflops": flop / (ms * 1e-3) / 1e12, "power_samples": len(busy),
               "avg_w": avg, "max_w": max(busy) if busy else None, "avg_sclk_mhz": sum(clk) / len(clk) if clk else None}
        if avg:
            rec["joules_per_launch"] = avg * ms * 1e-3
            rec["dynamic_joules_per_launch"] = (avg - idle_w) * ms * 1e-3
            rec["pj_per_flop"] = (avg - idle_w) * ms * 1e-3 / flop * 1e12
        out["classes"][name] = rec
        print(f"{name:44s} {ms * 1e3:8.1f} us  {rec['tflops']:7.1f} TF  {avg or 0:7.1f} W  {rec['avg_sclk_mhz'] or 0:6.0f} MHz  {rec.get('joules_per_launch', 0) * 1e3:7.1f} mJ/launch", flush=True)
    # the step = 22 layers x 2 micro-batches of (qkv, attention, wo, wi, wo_mlp): energy budget per class
    per_step = {"gemm_qkv (EPI_QKV_ROPE, N=2304 K=768)": 44, "gemm_wi (EPI_GEGLU, N=2304 K=768)": 44, "gemm_wo (EPI_RESIDUAL, N=768 K=768)": 44,
                "gemm_wo_mlp (EPI_RESIDUAL, N=768 K=1152)": 44, "attn_global (S=512)": 16, "attn_local (S=512, |i-j|<=64)": 28}
    # with the fused kernel (the default for this batch): 16 + 28 launches of it instead of 44 QKV GEMMs and 44 attention launches
    per_step_fused = {"qkv_attn fused global (S=512)": 16, "qkv_attn fused banded (S=512, |i-j|<=64)": 28, "gemm_wi (EPI_GEGLU, N=2304 K=768)": 44,
                      "gemm_wo (EPI_RESIDUAL, N=768 K=768)": 44, "gemm_wo_mlp (EPI_RESIDUAL, N=768 K=1152)": 44}
    if all(k in out["classes"] for k in per_step_fused):
        out["joules_per_step_sum_fused_schedule"] = sum(out["classes"][k].get("joules_per_launch", 0) * n for k, n in per_step_fused.items())
        out["ms_per_step_sum_isolated_fused_schedule"] = sum(out["classes"][k]["us_per_launch"] * n for k, n in per_step_fused.items()) / 1e3
    step = {k: out["classes"][k].get("joules_per_launch", 0) * n for k, n in per_step.items() if k in out["classes"]}
    out["joules_per_step_by_class"] = step
    out["joules_per_step_sum"] = sum(step.values())
    out["ms_per_step_sum_isolated"] = sum(out["classes"][k]["us_per_launch"] * n for k, n in per_step.items() if k in out["classes"]) / 1e3
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("idle_w", "joules_per_step_by_class", "joules_per_step_sum", "ms_per_step_sum_isolated")}))


if __name__ == "__main__":
    main()
