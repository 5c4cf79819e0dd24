#!/bin/bash
# One GPU session = one gpurun call (box time is budgeted per round): runs the named steps in order and leaves
# everything under gpurun_out/<tag>/ (copy what should be judged into profiles/).
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh r3a tests smoke bench energy'
# steps: tests | smoke | bench | bench_f16 | bench_large | energy | profile (kernel trace + PMC passes, tools/profile_round.sh)
#        | gemm (tools/gemm_bench.py 65536) | topk (tools/bench_topk.py, bench_dense_f32.py) | e2e | latency | <any other word>: run as a command
TAG=${1:-session}; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
for step in "$@"; do
  echo "== $step $(date +%T)" | tee -a "$OUT/steps.log"
  case "$step" in
    tests) python -m pytest tests -m gpu -q -x --durations=12 > "$OUT/pytest.log" 2>&1; tail -25 "$OUT/pytest.log" ;;
    smoke) python __graft_entry__.py smoke 2>&1 | tail -2 | tee "$OUT/smoke.log" ;;
    bench) python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -c 1500 "$OUT/bench.json" ;;
    bench_f16) python bench.py --operand-dtype f16 --cpu-budget 0 > "$OUT/bench_f16.json" 2>> "$OUT/bench.err"; cut -c1-300 "$OUT/bench_f16.json" ;;
    bench_large) python bench.py --model large --cpu-budget 0 > "$OUT/bench_large.json" 2>> "$OUT/bench.err"; cut -c1-300 "$OUT/bench_large.json" ;;
    energy) python tools/energy_by_class.py --out "$OUT/energy_by_class.json" 2>&1 | tee "$OUT/energy.log" ;;
    energy_step) python tools/energy_probe.py --steps 300 > "$OUT/energy_line.json" 2>&1; cat "$OUT/energy_line.json" ;;
    profile) bash tools/profile_round.sh "$TAG/prof" ;;
    gemm) python tools/gemm_bench.py 65536 2>&1 | tee "$OUT/gemm_bench.txt" ;;
    topk) python tools/bench_topk.py > "$OUT/topk_lines.json" 2>&1; python tools/bench_dense_f32.py > "$OUT/dense_f32_lines.json" 2>&1; tail -5 "$OUT/topk_lines.json" ;;
    e2e) python tools/bench_e2e.py > "$OUT/e2e_lines.json" 2>&1; tail -3 "$OUT/e2e_lines.json" ;;
    latency) python tools/bench_extract_latency.py 2>&1 | tee "$OUT/latency.txt" ;;
    *) bash -c "$step" 2>&1 | tee -a "$OUT/custom.log" ;;
  esac
done
