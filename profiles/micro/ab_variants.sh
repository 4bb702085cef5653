#!/bin/bash
# Same-box A/B of two builds of the library over the three variants of the north-star query (bench.py, kernels above 0.03 ms):
#   bash profiles/micro/ab_variants.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; R=${3:-2}
for round in $(seq 1 $R); do
	for L in "$A" "$B"; do
		for v in D U S; do
			MDB_LIBRARY=$L python bench.py --variant $v --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys,json
s=json.loads(sys.stdin.readlines()[-1]); print('$(basename $L)', '$v', round(s['ms_per_step'],4), {k:round(v['ms_per_step'],4) for k,v in s['kernels'].items() if v['ms_per_step']>0.03})"
		done
	done
done
