# developer check: the kbrl sub-record of the bench line (config 3: agents in the loop)
timeout 600 python bench.py --steps 100 --warmup 10 --burn-in 400 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kbrl']
print(json.dumps({a:k[a] for a in k if a not in ('workload',)}, indent=None)[:1500])"
