#!/bin/bash
# A/B of the leaf kernel with 64-bit hashes (MDB_NARROW_KEYS=0) and the narrow form (=1): SQ and LDS counters.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash profiles/micro/narrow_ab.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/narrow_ab
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
for m in 0 1; do
	export MDB_NARROW_KEYS=$m
	rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES \
		--output-format csv -d "$OUT/sq$m" -- python3 "$R/bench.py" $ARGS > "$OUT/sq$m.log" 2>&1
	rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
		--output-format csv -d "$OUT/lds$m" -- python3 "$R/bench.py" $ARGS > "$OUT/lds$m.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
for m in (0, 1):
    acc = collections.defaultdict(list)
    for sub in ("sq", "lds"):
        for path in glob.glob("%s/%s%d/*/*_counter_collection.csv" % (root, sub, m)):
            for r in csv.DictReader(open(path)):
                if "k_leaf_group_count" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("mode", m, {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
