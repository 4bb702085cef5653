#!/usr/bin/env python3
"""Which kernels (runtime copy / fill kernels included) run INSIDE one step of a traced program: rocprofv3 --kernel-trace lists every
dispatch with its start time; a step is delimited by successive launches of a marker kernel that runs once per step
(k_status_words for the sharded operator).  Usage: trace_per_step.py <dir with *_kernel_trace.csv> <marker substring> [out.json]"""
import collections
import csv
import glob
import json
import sys


def main():
    root, marker = sys.argv[1], sys.argv[2]
    rows = []
    for path in glob.glob(root + "/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 3:
        print("marker seen", len(marks), "times")
        return
    steps = []
    for a, b in zip(marks[:-1], marks[1:]):
        c = collections.Counter()
        t = collections.Counter()
        for s, e, k in rows[a + 1:b + 1]:
            k = k.split("(")[0].replace("void ", "")[:60]
            c[k] += 1
            t[k] += e - s
        steps.append({"wall_us": (rows[b][1] - rows[a][1]) / 1e3, "launches": dict(c), "busy_us": {k: v / 1e3 for k, v in t.items()}})
    last = steps[-1]
    print(f"{len(steps)} steps; the last one: {last['wall_us']:.1f} us between markers")
    for k, v in sorted(last["launches"].items(), key=lambda kv: -last["busy_us"][kv[0]]):
        print(f"  {v:4d} x {k:60s} {last['busy_us'][k]:9.1f} us")
    if len(sys.argv) > 3:
        json.dump({"marker": marker, "steps": steps[-4:]}, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
