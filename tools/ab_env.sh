# A/B inside one GPU session: every argument is an environment assignment list ("VAR=1 OTHER=2", "" = defaults) for one bench run
for cfg in "$@"; do echo "env: $cfg"; env $cfg VRAG_BENCH_SKIP_LEGS=1 timeout 300 python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['roofline'].get('isolated_pass',{}).get('classes',{})
print(round(d['value'],1), round(d['ms_per_step'],2), {k.split('::')[1][:22]: round(v['avg_launch_ms']*1e3,1) for k,v in c.items()})"; done
