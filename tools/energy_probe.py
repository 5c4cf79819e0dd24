#!/usr/bin/env python3
"""Energy per chunk: runs bench.py for a long timed region, samples `rocm-smi --showpower` meanwhile and prints the
average package power over the busy samples, joules per step and per chunk.  On this power-capped part
kernel variants should be compared by J/chunk, not only by microseconds: at the cap, throughput = (cap - static) / J.

  python tools/energy_probe.py [--steps 400] [--interval 0.5] [-- extra bench.py args / env via the shell]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

POWER_RE = re.compile(r"Current Socket Graphics Package Power \(W\):\s*([0-9.]+)")
CAP_RE = re.compile(r"Max Graphics Package Power \(W\):\s*([0-9.]+)")


def parse_power(text: str):
    m = POWER_RE.search(text)
    return float(m.group(1)) if m else None


def read_smi(args):
    try:
        return subprocess.run(["rocm-smi", *args], capture_output=True, text=True, timeout=10).stdout
    except Exception:
        return ""


def summarize(samples, idle_w, bench_line, cap_w=None):
    busy = [w for w in samples if w > idle_w + 0.5 * (max(samples) - idle_w)] if samples else []
    out = {"samples": len(samples), "busy_samples": len(busy), "idle_w": idle_w, "cap_w": cap_w}
    if busy and bench_line:
        avg = sum(busy) / len(busy)
        chunks_per_step = bench_line["config"].get("chunks_per_gpu_per_step", 256) * bench_line.get("n_gpus", 1)
        j_step = avg * bench_line["ms_per_step"] / 1e3
        out.update({"avg_busy_w": avg, "min_busy_w": min(busy), "max_busy_w": max(busy), "ms_per_step": bench_line["ms_per_step"],
                    "chunks_per_s": bench_line["value"], "joules_per_step": j_step, "joules_per_chunk": j_step / chunks_per_step,
                    "dynamic_joules_per_chunk": (avg - idle_w) * bench_line["ms_per_step"] / 1e3 / chunks_per_step})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--interval", type=float, default=0.5)
    ap.add_argument("bench_args", nargs="*", help="extra arguments for bench.py")
    args = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    idle_text = read_smi(["--showpower", "--showmaxpower"])
    idle_w = parse_power(idle_text) or 0.0
    cap = CAP_RE.search(idle_text)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--cpu-budget", "0", "--no-profile", "--steps", str(args.steps),
           "--warmup", str(args.warmup), *args.bench_args]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=root)
    samples = []
    while proc.poll() is None:
        w = parse_power(read_smi(["--showpower"]))
        if w is not None:
            samples.append(w)
        time.sleep(args.interval)
    line = None
    for ln in proc.stdout.read().splitlines():
        if ln.startswith("{"):
            line = json.loads(ln)
    print(json.dumps(summarize(samples, idle_w, line, float(cap.group(1)) if cap else None)))


if __name__ == "__main__":
    main()
