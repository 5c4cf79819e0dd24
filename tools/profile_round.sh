#!/bin/bash
# Collects the per-round evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01_v3
# -> gpurun_out/<tag>/{bench_line.json, kernel_stats.txt, pmc.txt}; copy what should be judged into profiles/.
# PMC passes are separate runs without any trace flag (gpurun refuses --pmc combined with tracing).
TAG=${1:-round}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --cpu-budget 0"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -- $B --steps 3 --warmup 1 > "$OUT/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -- $B --steps 2 --warmup 1 --no-profile > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -- $B --steps 2 --warmup 1 --no-profile > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  -d "$OUT/pmc_sq" -- $B --steps 2 --warmup 1 --no-profile > "$OUT/pmc_sq.log" 2>&1
cd "$REPO"
db() { find "$OUT/$1" -name '*.db' | head -1; }
python tools/rocpd_summary.py "$(db kt)" 1 3 > "$OUT/kernel_stats.txt" 2>&1
: > "$OUT/pmc.txt"
for d in pmc_fetch pmc_write pmc_sq; do python tools/pmc_summary.py "$(db $d)" >> "$OUT/pmc.txt" 2>&1; done
python tools/pmc_traffic.py "$OUT/pmc.txt" "$OUT/bench_line.json" > "$OUT/pmc_traffic.json" 2> "$OUT/pmc_traffic.err"
rm -rf "$OUT/kt" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq"
tail -c 600 "$OUT/bench_line.json"; head -20 "$OUT/kernel_stats.txt"
