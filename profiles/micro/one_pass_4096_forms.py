"""Which form the ordered operator takes for the shapes of tests/test_one_pass_4096_gpu.py (MDB_WIDE12_MIN=1)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from midoridb_amd.dev import DeviceCtx
os.environ["MDB_WIDE12_MIN"] = "1"
dev = DeviceCtx(0)
for bits in (24, 25, 26, 27):
    rng = np.random.default_rng(bits)
    span = (1 << bits) - 1000
    n_l, n_r = 1_200_000, 1_000_000
    kl = rng.permutation(span)[:n_l].astype(np.int64); kr = rng.permutation(span)[:n_r].astype(np.int64)
    kl[0], kl[-1] = 0, span - 1
    for r in range(2):
        k, c, f, j = dev.join_group_count(dev.to_dev(kl), None, dev.to_dev(kr), None)
        print(bits, r, "form", dev.last_join_form(), "one-pass", dev.last_join_one_pass_4096(), "groups", k.numel(), flush=True)
