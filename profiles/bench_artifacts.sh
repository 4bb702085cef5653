#!/bin/bash
# The bench lines kept under profiles/<tag>/ (run on the GPU box): bash profiles/bench_artifacts.sh r04
TAG=${1:-r04}; O=gpurun_out/bench_$TAG; mkdir -p $O
last() { tail -n 1; }
for V in D U S; do
	python bench.py --variant $V --steps 20 --warmup 5 2>/dev/null | last > $O/bench_variant$V.json
	python bench.py --variant $V --steps 20 --warmup 5 --force-shuffle --no-cpu-baseline 2>/dev/null | last > $O/bench_variant${V}_shuffle.json
done
python bench.py --config 4 --steps 10 --warmup 3 2>/dev/null | last > $O/bench_config4_local.json
python bench.py --config 4 --steps 10 --warmup 3 --reference-order 2>/dev/null | last > $O/bench_config4_local_reference_order.json
python bench.py --config 4 --steps 10 --warmup 3 --force-shuffle 2>/dev/null | last > $O/bench_config4_forced_shuffle.json
python bench.py --config 5 --steps 10 --warmup 3 2>/dev/null | last > $O/bench_config5_local.json
python bench.py --config 5 --steps 10 --warmup 3 --force-shuffle 2>/dev/null | last > $O/bench_config5_forced_shuffle.json
for f in $O/*.json; do python3 -c "
import json,sys
d=json.load(open('$f')); print('$(basename $f)', d.get('ms_per_step'), d.get('value'))"; done
