"""mdb_dev_group_count (first row + COUNT per key, first-row order) at 10^8 rows: through the tile sort (round 5) against the partitioned
path (MDB_GROUP_TILED=0), for 6.25M groups of 16 rows (bench_operators.py's workload) and for nearly unique keys:
    python profiles/micro/group_tiled.py [rows]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from midoridb_amd.dev import DeviceCtx

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
dev = DeviceCtx(0)
for name, keys in (("groups_of_16", dev.gen_keys(n, 0, n, 43, n // 16)), ("unique", dev.gen_keys(n, 0, n, 44, 0)),
                   ("groups_of_4_spread", dev.gen_keys(n, 0, n, 45, n // 4).mul_(4))):
    row = {"rows": n, "keys": name}
    ref = None
    for form in ("0", "1", "banded"):
        os.environ["MDB_GROUP_TILED"] = "1" if form == "1" else "0"
        os.environ["MDB_GROUP_BANDED"] = "1" if form == "banded" else "0"
        for _ in range(3):
            f, c = dev.group_count(keys, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            f, c = dev.group_count(keys, None)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        dev.prof_enable(True)
        dev.prof_reset()
        dev.group_count(keys, None)
        kern = {k: round(v[1], 4) for k, v in dev.prof_read().items() if v[1] > 0}
        dev.prof_enable(False)
        tag = {"0": "partitioned", "1": "tiled", "banded": "banded"}[form]
        row[tag + "_ms"], row[tag + "_kernels"], row["groups"] = round(ms, 4), kern, f.numel()
        if ref is None:
            ref = (f.clone(), c.clone())
        else:
            row["identical"] = bool(row.get("identical", True) and torch.equal(ref[0], f) and torch.equal(ref[1], c))
    print(json.dumps(row), flush=True)
    del keys, ref
