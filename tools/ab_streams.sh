# stream / micro-batch sweep of the bench step inside one GPU session: "ENV=..|bench args" per run
for cfg in "$@"; do envs=${cfg%%|*}; args=${cfg##*|}; echo "env: $envs args: $args"; env $envs VRAG_BENCH_SKIP_LEGS=1 timeout 300 python bench.py --steps 10 --warmup 3 $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2))"; done
