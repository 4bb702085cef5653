"""The fused join + GROUP BY + COUNT operator beyond 7 * 10^8 rows per table on ONE GPU (compact narrow form: direct-address
leaves have no table to overflow): 8 * 10^8 and 10^9 rows per table of the benchmark's generator, variant D; checks the
size-independent properties (groups, joined rows, sum of counts, every count = 16, keys unique) and times a step."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
for n in (800_000_000, 1_000_000_000):
    kl = dev.gen_keys(n, 0, n, 42, 0)
    kr = dev.gen_keys(n, 0, n, 43, n // 16)
    k, c, f, j = dev.join_group_count(kl, None, kr, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    k, c, f, j = dev.join_group_count(kl, None, kr, None)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    ok = (k.numel() == n // 16 and j == n and int(c.sum()) == n and int(c.min()) == 16 and int(c.max()) == 16 and
          int(torch.unique(k).numel()) == k.numel() and bool((f[1:].to(torch.int64) > f[:-1].to(torch.int64)).all()))
    print(n, "form", dev.last_join_form(), "filter", dev.last_join_filter(), "ms %.2f" % ms, "joined rows/s %.3g" % (j / ms * 1e3), "ok", ok, flush=True)
    del kl, kr, k, c, f
    torch.cuda.empty_cache()
