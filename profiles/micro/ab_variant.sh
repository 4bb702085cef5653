#!/bin/bash
# Same-box A/B of two builds of the library on one variant of the north-star query: bash profiles/micro/ab_variant.sh <libA.so> <libB.so> <D|U|S> [rounds]
A=$1; B=$2; V=$3; R=${4:-3}
for round in $(seq 1 $R); do
	for L in "$A" "$B"; do
		MDB_LIBRARY=$L python bench.py --variant $V --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys,json
s=json.loads(sys.stdin.readlines()[-1]); print('$(basename $L)', '$V', round(s['ms_per_step'],4), {k:round(v['ms_per_step'],4) for k,v in s['kernels'].items() if v['ms_per_step']>0.03})"
	done
done
