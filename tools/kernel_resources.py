#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py verbatim-rag_amd/csrc/gemm_bf16.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage", *sys.argv[2:]]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
demangle = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'spill':>6} {'scratch':>8} {'occ':>4}  kernel")
for r, d in zip(rows, demangle):
    print(f"{r.get('VGPRs','?'):>5} {r.get('AGPRs','?'):>5} {r.get('SGPRs','?'):>5} {r.get('VGPRs Spill','?'):>6} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>8} {r.get('Occupancy [waves/SIMD]','?'):>4}  {d[:150]}")
