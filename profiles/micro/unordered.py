"""join + GROUP BY + COUNT without MDB_ORDER_FIRST, 10^8 rows per table, variant U / S / D: run under rocprofv3 --kernel-trace --stats
for the per-kernel split (python3 profiles/micro/unordered.py U 8)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from midoridb_amd.dev import DeviceCtx

variant = sys.argv[1] if len(sys.argv) > 1 else "U"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000_000
dev = DeviceCtx(0)
a = dev.gen_keys(n, 0, n, 42, 0)
b = dev.gen_keys(n, 0, n, 43, 0 if variant == "U" else n // 16)
if variant == "S":
    b.mul_(16)
out = (torch.empty(n, dtype=torch.int64, device=dev.device), torch.empty(n, dtype=torch.int64, device=dev.device))
for _ in range(2):
    k, c, j = dev.join_group_count_unordered(a, None, b, None, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    k, c, j = dev.join_group_count_unordered(a, None, b, None, out=out)
torch.cuda.synchronize()
print(variant, "ms", (time.perf_counter() - t0) / reps * 1e3, "groups", k.numel(), "joined", j, "unordered form", dev.last_join_unordered())
