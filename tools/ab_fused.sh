# A/B of the fused QKV + attention kernel inside the bench step (one GPU session): "fused dbg" pairs from the command line,
# default: unfused, fused, fused without its attention phase (timing only: results are wrong with debug flags)
cfgs=("$@"); [ ${#cfgs[@]} -eq 0 ] && cfgs=("0 0" "2 0")
for cfg in "${cfgs[@]}"; do set -- $cfg; echo "fused=$1 dbg=$2"; VRAG_FUSED_QKV_ATTN=$1 VRAG_FUSED_DEBUG=$2 VRAG_BENCH_SKIP_LEGS=1 timeout 300 python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['roofline'].get('isolated_pass',{}).get('classes',{})
print(round(d['value'],1), round(d['ms_per_step'],2), {k.split('<')[1][:12]: round(v['avg_launch_ms']*1e3,1) for k,v in c.items()})"; done
