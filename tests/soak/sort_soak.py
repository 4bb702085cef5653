"""Randomised soak of mdb_dev_sort_perm against the numpy oracle: sizes around and above the packed path's threshold,
1-4 keys with random ranges / offsets / NULL rates / directions, occasional row-id vectors and clustered values."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import np_oracle as orc
from midoridb_amd import dev as D
dev = D.DeviceCtx(0)
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
t0 = time.time()
for seed in range(seeds):
    rng = np.random.default_rng(90_000 + seed)
    n = int(rng.choice([262_143, 262_144, 262_145, 500_000, 1_000_003, 2_500_000, int(10 ** rng.uniform(5.4, 6.6))]))
    nk = int(rng.integers(1, 5))
    with_rid = rng.random() < 0.25
    m = n if not with_rid else n + int(rng.integers(1, n))
    rid = rng.integers(0, m, n).astype(np.uint32) if with_rid else None
    rid_dev = dev.to_dev(rid) if with_rid else None
    keys_np, keys_dev, keep = [], [], []
    for c in range(nk):
        bits = int(rng.integers(1, 40 if nk == 1 else 18))
        off = int(rng.integers(-2**40, 2**40))
        shape = rng.random()
        if shape < 0.7:
            v = off + rng.integers(0, 2**bits, m, dtype=np.int64)
        elif shape < 0.85:
            v = off + (rng.integers(0, 2**bits, m, dtype=np.int64) >> int(rng.integers(0, bits))) * 3      # many duplicates
        else:
            v = off + np.floor(np.abs(rng.normal(0, 2 ** (bits / 2), m))).astype(np.int64)               # clustered near the offset
        nf = float(rng.choice([0.0, 0.0, 0.05, 0.5]))
        nulls = (rng.random(m) < nf) if nf else None
        desc = bool(rng.random() < 0.5)
        vd, nd = dev.to_dev(v), dev.nullbits_dev(nulls)
        keep += [vd, nd]
        keys_np.append((v, nulls, rid, False, desc))
        keys_dev.append((vd, nd, rid_dev, D.T_INT64, desc))
    got = dev.sort_perm(keys_dev, n).cpu().numpy().view(np.uint32)
    want = orc.sort_perm(keys_np, n)
    ok = np.array_equal(got, want)
    bad += not ok
    if not ok:
        print("MISMATCH seed", seed, "n", n, "keys", nk)
print(f"{seeds} cases, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
