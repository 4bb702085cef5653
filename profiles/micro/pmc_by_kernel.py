"""sum rocprofv3 --pmc counters per kernel name: python3 pmc_by_kernel.py <dir> [substring]"""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            calls[k] += 1
for k, c in acc.items():
    print(k[:90], "calls", calls[k])
    for n, v in sorted(c.items()):
        print("   %-28s %.4g per call" % (n, v / max(calls[k], 1)))
