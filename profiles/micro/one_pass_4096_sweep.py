"""Ordered join + GROUP BY + COUNT(*), unique keys on both sides (variant U's shape) at several table sizes: the one-pass 4096-digit form
(MDB_WIDE12_MIN=1) against the two-level form (MDB_WIDE12=0) - where the former starts to pay (mdb_dev_join.hip: gc_begin)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
for n in (8_000_000, 12_000_000, 16_000_000, 24_000_000, 33_000_000, 50_000_000, 67_000_000, 100_000_000):
    kl = dev.gen_keys(n, 0, n, 42, 0)
    kr = dev.gen_keys(n, 0, n, 43, 0)
    cap = n + 1024
    out = tuple(torch.empty(cap, dtype=torch.int64, device=dev.device) for _ in range(2))
    res = {}
    for mode in ("two-level", "one-pass"):
        os.environ.pop("MDB_WIDE12", None); os.environ.pop("MDB_WIDE12_MIN", None)
        if mode == "two-level":
            os.environ["MDB_WIDE12"] = "0"
        else:
            os.environ["MDB_WIDE12_MIN"] = "1"
        for _ in range(3):
            dev.join_group_count(kl, None, kr, None, out=out, want_first=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dev.join_group_count(kl, None, kr, None, out=out, want_first=False)
        torch.cuda.synchronize()
        res[mode] = ((time.perf_counter() - t0) / 10 * 1e3, dev.last_join_one_pass_4096(), dev.last_join_form())
    print(n, {k: (round(v[0], 3), v[1], v[2]) for k, v in res.items()}, flush=True)
    del kl, kr, out
    torch.cuda.empty_cache()
