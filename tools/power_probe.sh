#!/bin/bash
# Samples rocm-smi (package power, clocks) while bench.py runs a long timed region; one idle sample first.
# Output: gpurun_out/power_probe.txt (copy the summary to profiles/).
out=gpurun_out/power_probe.txt
{
echo "== idle"; rocm-smi --showpower --showmaxpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | head -12
python bench.py --cpu-budget 0 --no-profile --steps 700 --warmup 20 > gpurun_out/power_bench.json 2>/dev/null &
pid=$!
sleep 6
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14; do
  echo "== sample $i (bench running: $(kill -0 $pid 2>/dev/null && echo yes || echo no))"
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk" | head -6
  sleep 1.2
done
wait $pid
echo "== bench line"; cut -c1-330 gpurun_out/power_bench.json
} > $out 2>&1
cat $out
