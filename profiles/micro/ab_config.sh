#!/bin/bash
# Same-box A/B of two builds of the library over bench.py --config N (kernels above 0.03 ms; config 5: the join-only form too):
#   bash profiles/micro/ab_config.sh <libA.so> <libB.so> <config> [rounds] [extra bench.py flags]
A=$1; B=$2; C=$3; R=${4:-2}; shift 4 2>/dev/null
for round in $(seq 1 $R); do
	for L in "$A" "$B"; do
		MDB_LIBRARY=$L python bench.py --config $C --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys,json
s=json.loads(sys.stdin.readlines()[-1]); print('$(basename $L)', 'config $C', round(s['ms_per_step'],4), {k:round(v['ms_per_step'],4) for k,v in s.get('kernels',{}).items() if v['ms_per_step']>0.03})
j=s.get('join_only_form')
if j: print('   join-only', round(j['ms_per_step'],4), {k:round(v['ms_per_step'],4) for k,v in j['kernels'].items() if v['ms_per_step']>0.03})"
	done
done
