"""fact x dimension join: 10^8-row fact table, 10^6-row dimension whose keys are a subset of the fact keys' range"""
import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
n = 100_000_000
kl = dev.gen_keys(n, 0, n, 42, 0)
for nr, mod in ((1_000_000, 0), (10_000_000, 0), (100_000_000, n // 16)):
    kr = dev.gen_keys(nr, 0, n if mod else nr * 50, 43, mod)
    for mode in ("0", "1"):
        os.environ["MDB_SEMIJOIN"] = mode
        for _ in range(2):
            dev.join_group_count(kl, None, kr, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            k, c, f, j = dev.join_group_count(kl, None, kr, None)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        dev.prof_enable(True); dev.prof_reset(); dev.join_group_count(kl, None, kr, None); prof = dev.prof_read(); dev.prof_enable(False)
        print(nr, "semijoin", mode, "form", dev.last_join_form(), "filter", dev.last_join_filter(), "ms %.3f" % ms, "groups", k.numel(), "joined", j,
              {k2: round(v[1], 3) for k2, v in prof.items() if v[1] > 0.03}, flush=True)
