#!/bin/bash
# Same-box A/B of one environment knob on one variant of the north-star query: bash profiles/micro/ab_env.sh <VAR> <valueA> <valueB> <D|U|S> [rounds]
VAR=$1; A=$2; B=$3; V=$4; R=${5:-3}
for round in $(seq 1 $R); do
	for X in "$A" "$B"; do
		env $VAR=$X python bench.py --variant $V --steps 40 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys,json
s=json.loads(sys.stdin.readlines()[-1]); print('$VAR=$X', '$V', round(s['ms_per_step'],4), 'kernels', round(sum(v['ms_per_step'] for v in s['kernels'].values()),4))"
	done
done
