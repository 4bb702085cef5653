"""mdb_dev_join_group_count_multi at 10^8 unique keys per table (BASELINE configs[4] shape on one GPU): per-kernel times.
MDB_LD_REM=<bits> sizes the direct-address leaves."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from midoridb_amd.dev import DeviceCtx  # noqa: E402

dev = DeviceCtx(0)
n = 100_000_000
big = [dev.gen_keys(n, 0, n, s, 0) for s in (42, 43, 44)]
for it in range(5):
    torch.cuda.synchronize()
    t = time.perf_counter()
    k, c, f, j = dev.join_group_count_multi(big[0], None, [(big[1], None), (big[2], None)])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) * 1e3
print("ms", round(ms, 3), k.numel(), j, dev.last_join_multi())
dev.prof_enable(True)
dev.prof_reset()
dev.join_group_count_multi(big[0], None, [(big[1], None), (big[2], None)])
print({k_: round(v[1], 3) for k_, v in dev.prof_read().items() if v[1] > 0.05})
