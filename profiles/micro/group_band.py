"""Plain GROUP BY at 10^8 rows over 10^4 ... 10^6 distinct values in a dense or sparse window: which path, how long."""
import sys, time, torch
sys.path.insert(0, ".")
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
n = 100_000_000
torch.manual_seed(1)
case = 0
for spacing in (1, 3):
    for D in (5_000, 14_000, 33_000, 60_000, 130_000, 300_000, 1_000_000):
        case += 1       # (every case its own length: what the operator learned about a column is remembered by address and length)
        k = torch.randint(0, D, (n + 4096 * case,), dtype=torch.int64, device="cuda") * spacing - 7777
        for _ in range(3):
            f, c = dev.group_count(k, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            f, c = dev.group_count(k, None)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        dev.prof_enable(True); dev.prof_reset(); dev.group_count(k, None); prof = dev.prof_read(); dev.prof_enable(False)
        print("spacing", spacing, "D", D, "ms %.3f" % ms, "levels", dev.last_join_levels(), "groups", f.numel(),
              {k2: round(v[1], 3) for k2, v in prof.items() if v[1] > 0.05}, flush=True)
        del k
