#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 hit counters of the row-order join's kernels on BASELINE configs[4]'s join-only form:
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash profiles/micro/rj_pmc.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/rj_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
	T=$(echo $C | tr ' ' '_')
	rocprofv3 --pmc $C --output-format csv -d "$OUT/$T" -- python3 "$R/profiles/micro/config5_join_only_kernels.py" > "$OUT/$T.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        if "rj_" not in k and "gather" not in k:
            continue
        a = agg[k][row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
for k, cs in sorted(agg.items()):
    print(k, {c: (v[0], round(v[1] / v[0] / 1e6, 3)) for c, v in cs.items()}, "(launches, per launch in 1e6 units)")
PY
