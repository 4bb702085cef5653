#!/bin/bash
# Same-box A/B of one environment knob over the operator benchmarks: bash profiles/micro/ab_ops_env.sh <VAR> <valueA> <valueB>
for X in "$2" "$3" "$2" "$3"; do
	echo "== $1=$X"
	env $1=$X python bench_operators.py --out /tmp/ops.json > /dev/null 2>&1; python3 -c "
import json
d=json.load(open('/tmp/ops.json'))
def walk(x, p=''):
    if isinstance(x, dict):
        if 'ms' in x and isinstance(x['ms'], (int,float)): print('  ', p, round(x['ms'],4))
        else:
            for k,v in x.items(): walk(v, p+'/'+str(k))
walk(d)"
done
