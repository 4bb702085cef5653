for i in 1 2; do for X in 0 1; do MDB_PROBE=$X python bench.py --variant D --steps 40 --warmup 3 --no-secondary 2>/dev/null | python3 -c "
import sys,json
s=json.loads(sys.stdin.readlines()[-1]); print('MDB_PROBE=$X', round(s['ms_per_step'],4), {k: round(v['ms_per_step'],4) for k,v in s['kernels'].items()}, s.get('cpu_hash',{}).get('gpu_result_identical'))"; done; done
