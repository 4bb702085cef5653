#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 counters of the band-sorted GROUP BY's kernels (10^8 rows, 6.25 x 10^6 groups of 16):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash profiles/micro/bg_pmc.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/bg_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
	T=$(echo $C | tr ' ' '_')
	rocprofv3 --pmc $C --output-format csv -d "$OUT/$T" -- python3 "$R/profiles/micro/group_trace.py" run g16 > "$OUT/$T.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        if "gen_keys" in k or "rocclr" in k:
            continue
        a = agg[k][row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
calls = 4.0   # group_trace.py runs the operator four times
tot = 0.0
for k, cs in sorted(agg.items()):
    print(k, {c: (v[0], round(v[1] / v[0] / 1e6, 3)) for c, v in cs.items()}, "(launches, per launch in 1e6 units; FETCH_SIZE / WRITE_SIZE in KB)")
    tot += (2.0 * cs["FETCH_SIZE"][1] + cs["WRITE_SIZE"][1]) * 1024.0 / calls if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs else 0.0
print("HBM bytes per call (FETCH_SIZE x 2 (gfx950) + WRITE_SIZE, KB -> bytes): %.3f GB; algorithmic 8 B x 10^8 rows + 12 B x 6.25 x 10^6 groups = 0.875 GB" % (tot / 1e9))
PY
