#!/usr/bin/env python3
"""Derives HBM-side bytes per launch of the GEMM classes from a pmc_summary.py text (FETCH_SIZE and
WRITE_SIZE passes) and the bench line (for the micro-batch size).  FETCH_SIZE is doubled: on gfx950
rocprofv3 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section; checked
here on layernorm/embed kernels whose byte counts are known).  The counters sit on the L2's fabric
side, so Infinity-Cache hits are included: `hbm_bytes_per_launch` is an upper bound on DRAM traffic.
Usage: pmc_traffic.py pmc.txt bench_line.json > profiles/rNN_pmc_traffic.json"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import pretty  # noqa: E402


def parse(path):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = pretty(line.strip())
            out.setdefault(cur, {})
        else:
            m = re.match(r"\s+(\S+)\s+([0-9.]+) per dispatch", line)
            if m and cur:
                out[cur][m.group(1)] = float(m.group(2))
    return out


def main(pmc_path, bench_path):
    pmc = parse(pmc_path)
    bench = json.loads(open(bench_path).read().strip().splitlines()[-1])
    rows = min(bench["config"]["micro_batch_tokens"] or 131072, bench["config"]["chunks_per_gpu_per_step"] * bench["config"]["seq_len"])
    H, I = 768, 1152
    # residual GEMMs (round 4): the stream arrives and leaves as two 16-bit planes -- 4 B read + 4 B written per element, the
    # written high plane being the next GEMM's operand copy (round 3: 8 B of fp32 read-modify-write + a 2 B copy)
    stream = 2 * rows * H * 3   # operand plane + byte remainder plane, in and out (round 6; 8 B per element in rounds 4-5)
    alg = {
        "gemm_qkv": rows * H * 2 + 3 * H * H * 2 + 3 * rows * H * 2,
        "gemm_wi": rows * H * 2 + 2 * I * H * 2 + rows * I * 2,
        "gemm_wo": rows * H * 2 + H * H * 2 + stream,
        "gemm_wo_mlp": rows * I * 2 + H * I * 2 + stream,
    }
    epi = {"gemm_qkv": 5, "gemm_wi": 4, "gemm_wo": 3, "gemm_wo_mlp": 3}
    res = {}
    for cls, e in epi.items():
        name = next((k for k in pmc if f"gemm_bf16_kernel<{e}, 256, 256" in k and "FETCH_SIZE" in pmc[k]), None)
        if not name:
            continue
        f, w = pmc[name]["FETCH_SIZE"], pmc[name].get("WRITE_SIZE", 0.0)
        a = alg[cls] if e != 3 else (alg["gemm_wo"] + alg["gemm_wo_mlp"]) / 2
        res[cls] = {
            "kernel": re.search(r"gemm_bf16_kernel<[^>]*>", name).group(0), "rows_per_launch": rows,
            "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB": w,
            "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024,
            "algorithmic_bytes_per_launch": a,
            "note": "FETCH_SIZE x 2 (gfx950 wide-read correction) + WRITE_SIZE, separate --pmc passes; L2-fabric side, "
                    "Infinity-Cache hits included; kernel<3> = mean over the Wo and mlp-Wo launches",
        }
    for cls, tag in (("qkv_attn_global", "qkv_attn_kernel<false"), ("qkv_attn_local", "qkv_attn_kernel<true")):
        name = next((k for k in pmc if tag in k and "FETCH_SIZE" in pmc[k]), None)
        if not name:
            continue
        f, w = pmc[name]["FETCH_SIZE"], pmc[name].get("WRITE_SIZE", 0.0)
        res[cls] = {
            "kernel": name, "rows_per_launch": rows, "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB": w,
            "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024,
            "algorithmic_bytes_per_launch": rows * H * 2 + 3 * H * H * 2 + rows * H * 2,   # token rows in, weights, attention rows out
            "note": "fused Wqkv + RoPE + attention kernel: Q / K / V^T never reach the fabric; FETCH_SIZE x 2 + WRITE_SIZE as above",
        }
    import hashlib

    h = hashlib.sha256()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in ("gemm_bf16.hip", "gemm_bf16.h", "common.h", "qkv_attn.hip", "qkv_attn.h"):      # bench.py refuses the summary once these change
        with open(os.path.join(root, "verbatim-rag_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    res["_gemm_source_sha16"] = h.hexdigest()[:16]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
