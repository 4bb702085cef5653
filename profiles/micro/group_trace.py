"""One mdb_dev_group_count call (g16 | unique | g4) or one headline join + GROUP BY (joinD) at 10^8 rows under `rocprofv3 --kernel-trace`: the launch sequence of the LAST call (after warm-up), in order,
with durations and the gaps between kernels.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gtrace -- python3 profiles/micro/group_trace.py run
    python3 profiles/micro/group_trace.py show gpurun_out/gtrace"""
import glob, os, sys, csv
if sys.argv[1] == "run":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import torch
    from midoridb_amd.dev import DeviceCtx
    dev = DeviceCtx(0)
    n = 100_000_000
    shape = sys.argv[2] if len(sys.argv) > 2 else "g16"
    if shape == "joinD64":    # the headline's tables through the 64-bit form (keys that fit no 2^32 window get it)
        dev.lib.mdb_dev_set_narrow_keys(dev.h, 0)
        shape = "joinD"
    if shape == "joinD":      # the headline: A(10^8 unique keys) JOIN B(10^8 rows, 16 per key) GROUP BY key, COUNT(*)
        a, b = dev.gen_keys(n, 0, n, 42, 0), dev.gen_keys(n, 0, n, 43, n // 16)
        for _ in range(5):
            k, c, f, j = dev.join_group_count(a, None, b, None)
            torch.cuda.synchronize()
        print("groups", k.numel(), dev.last_plan())
        sys.exit(0)
    keys = dev.gen_keys(n, 0, n, 43, {"g16": n // 16, "unique": 0, "g4": n // 4}[shape])
    for _ in range(4):
        f, c = dev.group_count(keys, None)
        torch.cuda.synchronize()
    print("groups", f.numel(), dev.last_plan())
else:
    rows = []
    for fn in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        with open(fn) as f:
            rows += list(csv.DictReader(f))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last call = after the last gap of > 2 ms ... simpler: split at k_gen / large gaps
    starts = [int(r["Start_Timestamp"]) for r in rows]
    ends = [int(r["End_Timestamp"]) for r in rows]
    cut = 0
    for i in range(1, len(rows)):
        if "gen_keys" in rows[i - 1]["Kernel_Name"] or starts[i] - ends[i - 1] > 150_000:
            cut = i
    t0 = starts[cut]
    tot = 0
    for i in range(cut, len(rows)):
        d = ends[i] - starts[i]
        tot += d
        gap = starts[i] - ends[i - 1] if i > cut else 0
        print(f"{(starts[i] - t0) / 1e3:9.1f} us  {d / 1e3:8.1f} us  gap {gap / 1e3:6.1f}  {rows[i]['Kernel_Name'][:90]}")
    print(f"kernels {len(rows) - cut}, busy {tot / 1e3:.1f} us, span {(ends[-1] - t0) / 1e3:.1f} us")
