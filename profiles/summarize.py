#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by profiles/collect.sh into one JSON (per-kernel averages).

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB.  On gfx950 FETCH_SIZE counts exactly half of
the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section): `hbm_read_bytes`
below is FETCH_SIZE * 1024 * 2 and is only meaningful for the streaming kernels; `hbm_write_bytes`
is WRITE_SIZE * 1024.
"""
import collections
import csv
import glob
import json
import sys


def pmc(dirname):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(dirname + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def short(name):
    return name.replace("void ", "").split("(")[0]


def main():
    root, out = sys.argv[1], sys.argv[2]
    res = collections.defaultdict(dict)
    for path in glob.glob(root + "/kt/*/*_kernel_stats.csv"):
        for r in csv.DictReader(open(path)):
            k = short(r["Name"])
            res[k].update({"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"]),
                           "pct": float(r["Percentage"])})
    for sub, keep in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("sq", None), ("lds", None)):
        for k, d in pmc(root + "/" + sub).items():
            for c, v in d.items():
                if keep is None or c in keep:
                    res[short(k)][c] = v
    for k, d in res.items():
        if "FETCH_SIZE" in d:
            d["hbm_read_bytes"] = d["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in d:
            d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
    json.dump({"note": __doc__, "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
    for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("total_ns", 0))[:14]:
        print("%-44s calls %4d avg %9.1f us  read %8.1f MB write %8.1f MB" % (
            k[:44], d.get("calls", 0), d.get("avg_ns", 0) / 1e3, d.get("hbm_read_bytes", 0) / 1e6, d.get("hbm_write_bytes", 0) / 1e6))


if __name__ == "__main__":
    main()
