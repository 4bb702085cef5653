#!/bin/bash
# Commands that produce the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash profiles/collect.sh r04'
# Counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains
# besides --kernel-trace are combined with --pmc).  The program after "--" is python3 itself.
# One set of passes per workload of the north-star query: variant D (the headline), U (unique keys), S (the headline's duplication
# spread over the whole key range) and D in the wide form (64-bit hashes: MDB_NARROW_KEYS=0 is read by the library at context
# creation) -> summary_<tag>[_U|_S|_wide].json; kernel names need no table: bench.py's line carries, per profiler name, the names
# rocprofv3 lists its kernels under (mdb_dev_prof_symbols).
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
passes() {	# $1 = output directory, $2... = program and arguments
	local OUT=$1; shift
	mkdir -p "$OUT"
	rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- "$@" > "$OUT/kt.log" 2>&1
	rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- "$@" > "$OUT/fetch.log" 2>&1
	rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- "$@" > "$OUT/write.log" 2>&1
	rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES \
		--output-format csv -d "$OUT/sq" -- "$@" > "$OUT/sq.log" 2>&1
	rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
		--output-format csv -d "$OUT/lds" -- "$@" > "$OUT/lds.log" 2>&1
}
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
for V in D U S; do
	SUF=""; [ "$V" != "D" ] && SUF="_$V"
	passes "$R/gpurun_out/prof_$TAG$SUF" python3 "$R/bench.py" $ARGS --variant $V
	(cd "$R" && python3 profiles/summarize.py "$R/gpurun_out/prof_$TAG$SUF" "$R/gpurun_out/summary_$TAG$SUF.json")
done
export MDB_NARROW_KEYS=0
passes "$R/gpurun_out/prof_${TAG}_wide" python3 "$R/bench.py" $ARGS --variant D
unset MDB_NARROW_KEYS
(cd "$R" && python3 profiles/summarize.py "$R/gpurun_out/prof_${TAG}_wide" "$R/gpurun_out/summary_${TAG}_wide.json")
# the sharded operator on one GPU (forced shuffle: partition by destination, RCCL all-to-all with itself, split local join)
passes "$R/gpurun_out/prof_${TAG}_shuffle" python3 "$R/bench.py" $ARGS --variant D --force-shuffle
(cd "$R" && python3 profiles/summarize.py "$R/gpurun_out/prof_${TAG}_shuffle" "$R/gpurun_out/summary_${TAG}_shuffle.json")
# BASELINE configs[0..1] shapes (scan + WHERE + projection at 10^8 rows, join with payload at 10^7): time + HBM bytes per kernel
OPS=$R/gpurun_out/prof_$TAG/configs1
mkdir -p "$OPS"
OARGS="--configs1 --out $OPS/operators.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OPS/kt" -- python3 "$R/bench_operators.py" $OARGS > "$OPS/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OPS/fetch" -- python3 "$R/bench_operators.py" $OARGS > "$OPS/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OPS/write" -- python3 "$R/bench_operators.py" $OARGS > "$OPS/write.log" 2>&1
(cd "$R" && python3 profiles/summarize.py "$OPS" "$R/gpurun_out/summary_${TAG}_configs1.json")
# the operator without MDB_ORDER_FIRST (groups in unspecified order: no row ids, no ordering sort), variants D and U
for V in D U; do
	passes "$R/gpurun_out/prof_${TAG}_unordered_$V" python3 "$R/bench.py" $ARGS --variant $V --unordered
	(cd "$R" && python3 profiles/summarize.py "$R/gpurun_out/prof_${TAG}_unordered_$V" "$R/gpurun_out/summary_${TAG}_unordered_$V.json")
done
# BASELINE configs[3] on one GPU (round 4): the join of two key columns, unique keys, 10^8 rows per table - mdb_dev_join_keys
# (any order, the sharded form's pipeline on local regions): time + HBM bytes per kernel
C4=$R/gpurun_out/prof_${TAG}_config4
mkdir -p "$C4"
C4ARGS="--config 4 --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d "$C4/kt" -- python3 "$R/bench.py" $C4ARGS > "$C4/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$C4/fetch" -- python3 "$R/bench.py" $C4ARGS > "$C4/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$C4/write" -- python3 "$R/bench.py" $C4ARGS > "$C4/write.log" 2>&1
(cd "$R" && python3 profiles/summarize.py "$C4" "$R/gpurun_out/summary_${TAG}_config4.json")
# round 5: the other BASELINE configurations through bench.py --config N (bench_configs.py reads these summaries for roofline.traffic):
# configs[1] (--config 2: join with payload at 10^7 rows through query_execute), configs[3] in the reference's row order, configs[4]
# (--config 5: the grouped statement and its join-only form in one run - the row-order payload join's kernels, mdb_dev_rowjoin.hip)
cfg_passes() {	# $1 = suffix, $2... = bench.py arguments
	local SUF=$1; shift
	local D=$R/gpurun_out/prof_${TAG}_$SUF
	mkdir -p "$D"
	rocprofv3 --kernel-trace --stats --output-format csv -d "$D/kt" -- python3 "$R/bench.py" "$@" --no-cpu-baseline > "$D/kt.log" 2>&1
	rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/fetch" -- python3 "$R/bench.py" "$@" --no-cpu-baseline > "$D/fetch.log" 2>&1
	rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/write" -- python3 "$R/bench.py" "$@" --no-cpu-baseline > "$D/write.log" 2>&1
	(cd "$R" && python3 profiles/summarize.py "$D" "$R/gpurun_out/summary_${TAG}_$SUF.json")
}
cfg_passes config2 --config 2 --steps 5 --warmup 2
cfg_passes config4_reference_order --config 4 --reference-order --steps 3 --warmup 1
cfg_passes config5 --config 5 --steps 4 --warmup 1
# the join-only statement of configs[4] ALONE (profiles/micro/config5_join_only_kernels.py): in the run above the grouped statement launches
# kernels of the same names, and traffic per launch averaged over both would price neither (advisor, round 5)
J5=$R/gpurun_out/prof_${TAG}_config5_join
mkdir -p "$J5"
rocprofv3 --kernel-trace --stats --output-format csv -d "$J5/kt" -- python3 "$R/profiles/micro/config5_join_only_kernels.py" > "$J5/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$J5/fetch" -- python3 "$R/profiles/micro/config5_join_only_kernels.py" > "$J5/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$J5/write" -- python3 "$R/profiles/micro/config5_join_only_kernels.py" > "$J5/write.log" 2>&1
(cd "$R" && python3 profiles/summarize.py "$J5" "$R/gpurun_out/summary_${TAG}_config5_join.json")
# what goes under profiles/$TAG/: the summaries and rocprofv3's own per-kernel statistics of each kernel-trace pass
PUB=$R/gpurun_out/publish_$TAG
mkdir -p "$PUB"
for S in "" _U _S _wide _shuffle _configs1 _unordered_D _unordered_U _config4 _config2 _config4_reference_order _config5 _config5_join; do
	[ -f "$R/gpurun_out/summary_$TAG$S.json" ] && cp "$R/gpurun_out/summary_$TAG$S.json" "$PUB/rocprof_summary$S.json"
	D="$R/gpurun_out/prof_$TAG$S/kt"
	[ "$S" = _configs1 ] && D="$OPS/kt"
	[ "$S" = _config4 ] && D="$C4/kt"
	[ "$S" = _config5_join ] && D="$J5/kt"
	F=$(find "$D" -name '*kernel_stats.csv' 2>/dev/null | head -1)
	[ -n "$F" ] && cp "$F" "$PUB/kernel_stats$S.csv"
done
