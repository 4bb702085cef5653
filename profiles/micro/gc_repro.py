"""Regression probe: plain GROUP BY over all-unique keys at sizes where the ordering sort's first-level digits
cover only part of their range (row ids < n < 2^kbits).  Prints which sort kernels ran: `sort_hist_l0` means
the fixed-capacity regions overflowed and the exact layout had to redo the sort."""
import sys, torch
sys.path.insert(0, '.')
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
for N in [1_000_000, 3_000_000, 20_000_000, 50_000_000, 67_108_864, 67_108_865, 100_000_000]:
    a = dev.gen_keys(N, 0, N, 42, 0)
    dev.prof_enable(True); dev.prof_reset()
    first, cnt = dev.group_count(a, None)
    prof = dev.prof_read(); dev.prof_enable(False)
    bad = int((cnt != 1).sum())
    fbad = int((first.long() != torch.arange(first.numel(), device=first.device)).sum())
    print(N, first.numel(), "bad counts", bad, "bad firsts", fbad, {k: round(v[1], 3) for k, v in prof.items() if 'sort' in k or 'order' in k})
    del a, first, cnt
