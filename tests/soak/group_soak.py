"""Randomised soak of plain GROUP BY + COUNT(*) (mdb_dev_group_count) against the numpy oracle: cardinalities from 1 to n,
values dense / scattered / clustered with outliers, NULLs, and ONE device buffer refilled in place between cases so that
remembered samples and verdicts (direct tables, hashed tables, narrow form) are stale as often as they are right."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from oracle import np_oracle as orc
from midoridb_amd import dev as D
dev = D.DeviceCtx(0)
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0
t0 = time.time()
sizes = [300_000, 1_500_000, 3_000_000, 4_700_001, 6_291_456]       # (from 2^21 rows on: the band sort, windows of 2^18 key values and more)
bufs = {n: torch.empty(n, dtype=torch.int64, device=dev.device) for n in sizes}
for seed in range(seeds):
    rng = np.random.default_rng(50_000 + seed)
    n = int(rng.choice(sizes))
    d = int(min(n, max(1, 10 ** rng.uniform(0, 6.5))))
    shape = rng.choice(["dense", "scattered", "offset", "outlier", "hot"])
    if shape == "dense":
        vals = np.arange(d, dtype=np.int64) - d // 3
    elif shape == "scattered":
        vals = np.unique(rng.integers(np.iinfo(np.int64).min, np.iinfo(np.int64).max, d, dtype=np.int64))
    elif shape == "offset":
        vals = 10**15 + np.arange(d, dtype=np.int64) * int(rng.choice([1, 3, 1000]))
    else:
        vals = np.arange(d, dtype=np.int64) * int(rng.choice([1, 1, 7]))
    k = vals[rng.integers(0, len(vals), n)]
    if shape == "hot":      # a run of one value: a digit's region of a band overflows (band sort), a leaf region overflows (partitioned path)
        a = int(rng.integers(0, n - n // 8))
        k[a:a + n // 8] = vals[0]
    if shape == "outlier":
        k[int(rng.integers(0, n))] = -2**40
        k[int(rng.integers(0, n))] = 2**50
    nulls = (rng.random(n) < float(rng.choice([0.01, 0.3]))) if rng.random() < 0.4 else None
    bufs[n].copy_(torch.from_numpy(k))          # same pointer as the previous case of this size
    first, cnt = dev.group_count(bufs[n], dev.nullbits_dev(nulls))
    e_first, e_cnt = orc.group_count(k, nulls)
    ok = np.array_equal(first.cpu().numpy().view(np.uint32).astype(np.int64), e_first) and np.array_equal(cnt.cpu().numpy(), e_cnt)
    bad += not ok
    if not ok:
        print("MISMATCH seed", seed, "n", n, "d", d, shape)
print(f"{seeds} cases, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
