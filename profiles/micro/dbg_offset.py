import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
shape = "offset"
rng = np.random.default_rng(len(shape) * 13 + 5)
n_l, n_r, span = 3_000_000, 2_500_000, 3_000_000
kl = rng.permutation(span)[:n_l].astype(np.int64)
off = -(2**45)
kr = rng.integers(0, span // 4, n_r, dtype=np.int64)
kl, kr = kl + off, kr + off
for mode in ("0", "1", "0", "1"):
    os.environ["MDB_SEMIJOIN"] = mode
    dl, dr = dev.to_dev(kl.copy()), dev.to_dev(kr.copy())
    k, c, f, j = dev.join_group_count(dl, None, dr, None)
    print("mode", mode, "form", dev.last_join_form(), "filter", dev.last_join_filter(), "groups", k.numel(), "joined", j, flush=True)
for off2 in (0, -(2**40), -(2**45), 2**45):
    os.environ["MDB_SEMIJOIN"] = "0"
    dl, dr = dev.to_dev(kl - off + off2), dev.to_dev(kr - off + off2)
    k, c, f, j = dev.join_group_count(dl, None, dr, None)
    print("off", off2, "form", dev.last_join_form(), flush=True)
