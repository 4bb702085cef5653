"""Same-process A/B of the row-order probe (mdb_dev_probe.hip, MDB_PROBE=0|1 alternating) on the headline's tables (variant D, 10^8 rows per table)
with the catalog's statistics handed over as query_execute() does: results compared element for element, time per step, kernels."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from midoridb_amd.dev import DeviceCtx

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = DeviceCtx(0)
a = dev.gen_keys(n, 0, n, 42, 0)
b = dev.gen_keys(n, 0, n, 43, n // 16)
sa = dev.key_range(a) + (1 if dev.distinct(a) else 0,)
sb = dev.key_range(b) + (0,)
print("statistics", sa, sb, flush=True)
res = {}
for rnd in range(3):
    for knob in ("0", "1"):
        os.environ["MDB_PROBE"] = knob
        dev.call_stats(a, sa, b, sb)
        for _ in range(3):
            k, c, f, j = dev.join_group_count(a, None, b, None)
        torch.cuda.synchronize()
        each = []
        for _ in range(20):
            t = time.perf_counter()
            dev.join_group_count(a, None, b, None, want_first=False)
            torch.cuda.synchronize()
            each.append((time.perf_counter() - t) * 1e3)
        plan = dev.last_plan()
        dev.prof_enable(True); dev.prof_reset()
        for _ in range(3):
            dev.join_group_count(a, None, b, None, want_first=False)
        prof = {kk: round(v[1] / 3, 4) for kk, v in dev.prof_read().items() if v[1] / 3 > 0.004}
        dev.prof_enable(False)
        dev.call_stats()
        each.sort()
        print(f"MDB_PROBE={knob}: min {each[0]:.4f} median {each[10]:.4f} max {each[-1]:.4f} ms, groups {k.numel()}, joined {j}, left_row_order {plan['left_row_order']}, "
              f"retries {plan['retries']}, kernels {prof}", flush=True)
        if knob in res:
            pass
        res.setdefault(knob, (k.clone(), c.clone(), f.clone(), j))
k0, c0, f0, j0 = res["0"]
k1, c1, f1, j1 = res["1"]
print("identical:", j0 == j1 and torch.equal(k0, k1) and torch.equal(c0, c1) and torch.equal(f0, f1))
