#!/usr/bin/env python3
"""Is an HBM-bound phase expensive in itself, or only because the chip idles at full clock while it waits?  (Round 3: every
kernel class of the step -- MFMA main loops and HBM-speed epilogues alike -- sits at the 1 400 W package cap,
profiles/r03_energy_by_class.json; the step is the SUM of 22 ms of main loops and 18 ms of HBM time.)  Three loops, package power
and clocks sampled meanwhile:
  (a) a device-to-device copy stream alone (pure HBM traffic, no MFMA),
  (b) the bare GEMM main loop alone (EPI_NONE, no HBM traffic to speak of),
  (c) both at once on two streams -- the rates each keeps tell whether overlapping the two kinds of phases can pay.
  python tools/power_overlap_probe.py [--seconds 5] > gpurun_out/power_overlap_probe.json
"""
import argparse
import ctypes as C
import json
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import verbatim_rag_amd  # noqa: E402,F401
from verbatim_rag_amd import _lib  # noqa: E402

POWER_RE = re.compile(r"Current Socket Graphics Package Power \(W\):\s*([0-9.]+)")
SCLK_RE = re.compile(r"sclk clock level:.*\((\d+)Mhz\)")


def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
    except Exception:
        return None, None
    p, c = POWER_RE.search(out), SCLK_RE.search(out)
    return (float(p.group(1)) if p else None), (int(c.group(1)) if c else None)


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], threading.Event()

    def run(self):
        while not self.stop.is_set():
            w, f = sample()
            if w is not None:
                self.rows.append((time.perf_counter(), w, f))
            self.stop.wait(0.05)


def measure(fn, seconds):
    smp = Sampler()
    t0 = time.perf_counter()
    smp.start()
    out = fn(seconds)
    t1 = time.perf_counter()
    smp.stop.set()
    smp.join()
    busy = [(w, f) for t, w, f in smp.rows if t0 + 0.3 * (t1 - t0) <= t <= t1]
    out["avg_w"] = sum(w for w, _ in busy) / len(busy) if busy else None
    fs = [f for _, f in busy if f]
    out["avg_sclk_mhz"] = sum(fs) / len(fs) if fs else None
    out["samples"] = len(busy)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=5.0)
    args = ap.parse_args()
    lib = _lib.load()
    dbg = _lib.load_debug()   # harness library (include/vrag_amd_debug.h)
    dev = torch.device("cuda", 0)
    n = 1 << 28                                            # 1 GiB of fp32 per buffer: far beyond the Infinity Cache
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    side = torch.cuda.Stream(device=dev)
    M, N, K = 65536, 2304, 768

    def copy_loop(seconds, stream=None):
        reps = 0
        with torch.cuda.stream(stream or torch.cuda.current_stream()):
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                for _ in range(20):
                    dst.copy_(src, non_blocking=True)
                reps += 20
                (stream or torch.cuda.current_stream()).synchronize()
            dt = time.perf_counter() - t0
        return {"copy_tb_per_s": reps * 2.0 * n * 4 / dt / 1e12}

    def gemm_loop(seconds):
        ms = C.c_float()
        _lib.check_debug("gemm", dbg.vrag_debug_gemm_ms(7, M, N, K, 50, 0, C.byref(ms)))
        iters = max(100, int(seconds / (ms.value * 1e-3)))
        _lib.check_debug("gemm", dbg.vrag_debug_gemm_ms(7, M, N, K, iters, 0, C.byref(ms)))
        return {"gemm_tflops": 2.0 * M * N * K / (ms.value * 1e-3) / 1e12, "us_per_launch": ms.value * 1e3}

    def both(seconds):
        box = {}
        th = threading.Thread(target=lambda: box.update(gemm_loop(seconds)))
        th.start()
        c = copy_loop(seconds + 0.5, side)
        th.join()
        return {**box, **c}

    idle = [sample() for _ in range(4) if time.sleep(0.2) is None]
    out = {"idle_w": sum(w for w, _ in idle if w) / max(1, len([1 for w, _ in idle if w]))}
    out["copy_alone"] = measure(lambda s: copy_loop(s), args.seconds)
    out["gemm_mainloop_alone"] = measure(gemm_loop, args.seconds)
    out["both_at_once"] = measure(both, args.seconds)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
