#!/bin/bash
# A/B of library builds inside one GPU session: alternating bench runs (headline + isolated class means), `reps` rounds.
#   tools/ab_libs.sh 2 "" base        ""  = the in-tree libvrag_amd.so, <tag> = libvrag_amd_<tag>.so (VRAG_BUILD_VARIANT=<tag>)
REPS=$1; shift
for r in $(seq 1 $REPS); do for tag in "$@"; do
  lib=verbatim-rag_amd/libvrag_amd${tag:+_$tag}.so; echo "## $lib"
  VRAG_AMD_LIB=$PWD/$lib VRAG_BENCH_SKIP_LEGS=1 timeout 300 python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['roofline'].get('isolated_pass',{}).get('classes',{})
print(round(d['value'],1), round(d['ms_per_step'],2), {k.split('::')[1][:22]: round(v['avg_launch_ms']*1e3,1) for k,v in c.items()}, 'parity', d.get('parity_max_abs_err_vs_oracle'))"
done; done
