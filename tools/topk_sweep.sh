#!/bin/bash
# tuning helper: dense single-query top-k time vs grid size (VRAG_TOPK_WGS)
for w in "$@"; do
  echo "wgs=$w"; VRAG_TOPK_WGS=$w python tools/bench_topk.py 2>&1 | grep '"nq": 1,' | head -1 | cut -c1-220
done
