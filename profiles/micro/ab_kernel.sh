#!/bin/bash
# Same-box A/B of one environment knob, printing the step time and the kernels whose name matches a pattern:
# bash profiles/micro/ab_kernel.sh <VAR> <valueA> <valueB> <D|U|S> <name pattern> [rounds]
VAR=$1; A=$2; B=$3; V=$4; PAT=$5; R=${6:-2}
for round in $(seq 1 $R); do
	for X in "$A" "$B"; do
		env $VAR=$X python bench.py --variant $V --steps 40 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys,json,re
s=json.loads(sys.stdin.readlines()[-1]); print('$VAR=$X', '$V', round(s['ms_per_step'],4), {k: round(v['ms_per_step'],4) for k,v in s['kernels'].items() if re.search('$PAT',k)})"
	done
done
