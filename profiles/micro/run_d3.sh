for i in 1 2 3; do python bench.py --variant D --steps 40 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys,json
s=json.loads(sys.stdin.readlines()[-1]); print(round(s['ms_per_step'],4), {k: round(v['ms_per_step'],4) for k,v in s['kernels'].items()})"; done
