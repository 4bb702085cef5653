#!/bin/bash
# Same-box A/B of two builds of the library over the operator benchmarks and the other configs:
#   bash profiles/micro/ab_ops.sh <libA.so> <libB.so>
for L in "$1" "$2"; do
	echo "== $(basename $L)"
	MDB_LIBRARY=$L python bench_operators.py --out /tmp/ops.json > /dev/null 2>&1; python3 -c "
import json
d=json.load(open('/tmp/ops.json'))
def walk(x, p=''):
    if isinstance(x, dict):
        if 'ms' in x and isinstance(x['ms'], (int,float)): print('  ', p, round(x['ms'],4))
        else:
            for k,v in x.items(): walk(v, p+'/'+str(k))
walk(d)"
	MDB_LIBRARY=$L python bench_operators.py --configs1 --out /tmp/ops1.json > /dev/null 2>&1; python3 -c "
import json
d=json.load(open('/tmp/ops1.json'))
def walk(x, p=''):
    if isinstance(x, dict):
        if 'ms' in x and isinstance(x['ms'], (int,float)): print('  c1', p, round(x['ms'],4))
        else:
            for k,v in x.items(): walk(v, p+'/'+str(k))
walk(d)"
	for C in 4 5; do MDB_LIBRARY=$L python bench.py --config $C --steps 6 --warmup 2 2>/dev/null | python3 -c "
import sys,json
s=json.loads(sys.stdin.readlines()[-1]); print('   config $C', round(s['ms_per_step'],4), round(s.get('join_only_form',{}).get('ms_per_step',0),3))"; done
done
