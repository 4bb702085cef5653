#!/bin/bash
# Commands that produce the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash profiles/collect.sh r01'
# Counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains
# besides --kernel-trace are combined with --pmc).  The program after "--" is python3 itself.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/bench.py" $ARGS > "$OUT/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$R/bench.py" $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$R/bench.py" $ARGS > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES \
	--output-format csv -d "$OUT/sq" -- python3 "$R/bench.py" $ARGS > "$OUT/sq.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
	--output-format csv -d "$OUT/lds" -- python3 "$R/bench.py" $ARGS > "$OUT/lds.log" 2>&1
cd "$R" && python3 profiles/summarize.py "$OUT" "$R/gpurun_out/summary_$TAG.json"
# BASELINE configs[0..1] shapes (scan + WHERE + projection at 10^8 rows, join with payload at 10^7): time + HBM bytes per kernel
OPS=$OUT/configs1
mkdir -p "$OPS"
cd /tmp
OARGS="--configs1 --out $OPS/operators.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OPS/kt" -- python3 "$R/bench_operators.py" $OARGS > "$OPS/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OPS/fetch" -- python3 "$R/bench_operators.py" $OARGS > "$OPS/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OPS/write" -- python3 "$R/bench_operators.py" $OARGS > "$OPS/write.log" 2>&1
cd "$R" && python3 profiles/summarize.py "$OPS" "$R/gpurun_out/summary_${TAG}_configs1.json"
