#!/bin/bash
# The bench lines kept under profiles/<tag>/ (run on the GPU box): bash profiles/bench_artifacts.sh r06
TAG=${1:-r06}; O=gpurun_out/bench_$TAG; mkdir -p $O
last() { tail -n 1; }
python bench.py 2>/dev/null | last > $O/bench_driver_command.json		# what the driver runs: N = 1, variant D, defaults
for V in U S; do
	python bench.py --variant $V --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | last > $O/bench_variant$V.json
done
for V in D U S; do
	python bench.py --variant $V --steps 20 --warmup 5 --force-shuffle --no-cpu-baseline 2>/dev/null | last > $O/bench_variant${V}_shuffle.json
done
python bench.py --config 2 --steps 10 --warmup 3 2>/dev/null | last > $O/bench_config2_local.json
python bench.py --config 4 --steps 10 --warmup 3 2>/dev/null | last > $O/bench_config4_local.json
python bench.py --config 4 --steps 10 --warmup 3 --reference-order 2>/dev/null | last > $O/bench_config4_local_reference_order.json
python bench.py --config 4 --steps 10 --warmup 3 --force-shuffle --no-cpu-baseline 2>/dev/null | last > $O/bench_config4_forced_shuffle.json
python bench.py --config 5 --steps 10 --warmup 3 2>/dev/null | last > $O/bench_config5_local.json
python bench.py --config 5 --steps 10 --warmup 3 --force-shuffle --no-cpu-baseline 2>/dev/null | last > $O/bench_config5_forced_shuffle.json
# the N > 1 code of bench.py on this one GPU (host-memory transport: correctness of the launcher path, not timings)
python bench.py --gpus 2 --transport test --rows 2000000 --steps 3 --warmup 1 --verify --no-cpu-baseline 2>/dev/null | last > $O/bench_world2_test_transport.json
python bench.py --gpus 8 --transport test --rows 1000000 --steps 3 --warmup 1 --verify --no-cpu-baseline 2>/dev/null | last > $O/bench_world8_test_transport.json
python bench_operators.py --out $O/operators.json > /dev/null 2>&1
python profiles/micro/join_payload_forms.py 2>/dev/null | grep '^{' > $O/join_payload_forms.json
for f in $O/bench_*.json; do python3 -c "
import json,sys
try:
    d=json.load(open('$f')); print('$(basename $f)', d.get('ms_per_step'), d.get('value'))
except Exception as e:
    print('$(basename $f)', 'unreadable:', e)"; done
