"""GROUP BY + COUNT(*) of one column whose 10^8 values are all distinct (and 2 rows per value): the plain GROUP BY's worst case - every row a group."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
for mod, what in ((0, "unique"), (n // 2, "2 rows per value")):
    k = dev.gen_keys(n, 0, n, 42, mod)
    for _ in range(3):
        first, cnt = dev.group_count(k, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        first, cnt = dev.group_count(k, None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    dev.prof_enable(True); dev.prof_reset(); dev.group_count(k, None); prof = dev.prof_read(); dev.prof_enable(False)
    print(what, "groups", cnt.numel(), round(ms, 3), "ms", {a: round(b[1], 3) for a, b in sorted(prof.items(), key=lambda kv: -kv[1][1])[:8]}, flush=True)
