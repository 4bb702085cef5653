"""Randomised LARGE shapes (3 * 10^7 ... 1.2 * 10^8 rows per table: first-level tiles of 8192 keys, the one-pass 4096-digit form, the one-level
pruned form with ranged ordering) against the multi-threaded C hash-join oracle - every call twice (the second runs on what the first learned);
then GROUP BY + COUNT(*) of the right table's column alone (band sort / tile sort / partitioned path by window), checked against torch on the device:
    python tests/soak/large_soak.py [cases]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from oracle import cpu
from midoridb_amd import dev as D
dev = D.DeviceCtx(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bad = 0
t0 = time.time()
for seed in range(cases):
    rng = np.random.default_rng(90_000 + seed)
    n_l = int(rng.integers(30_000_000, 120_000_000))
    n_r = int(rng.integers(20_000_000, 120_000_000))
    shape = int(rng.integers(0, 4))
    off = int(rng.choice([0, 12345, -(2**40), 10**12]))
    if shape == 0:      # dimension in the low part of the left table's range (variant D's shape)
        frac = int(rng.choice([8, 16, 32]))
        kl = rng.permutation(n_l).astype(np.int64)
        kr = rng.integers(0, max(n_l // frac, 1), n_r, dtype=np.int64)
    elif shape == 1:    # both sides over the same range, few duplicates (variant U's shape)
        span = int(max(n_l, n_r) * float(rng.choice([1.0, 1.2])))
        kl = rng.permutation(span)[:n_l].astype(np.int64)
        kr = rng.permutation(span)[:n_r].astype(np.int64)
    elif shape == 2:    # duplicates spread over the whole range (variant S's shape)
        dup = int(rng.choice([4, 16]))
        kl = rng.permutation(n_l).astype(np.int64)
        kr = (rng.integers(0, max(n_l // dup, 1), n_r, dtype=np.int64) * dup)
    else:               # duplicates on both sides, random keys
        span = int(2 ** rng.uniform(20, 27))
        kl = rng.integers(0, span, n_l, dtype=np.int64)
        kr = rng.integers(0, span, n_r, dtype=np.int64)
    kl += off
    kr += off
    ek, ec, ef, ej = cpu.hash_join_group_count(kl, None, kr, None, 64)
    dl, dr = dev.to_dev(kl), dev.to_dev(kr)
    for round_ in range(2):
        k, c, f, j = dev.join_group_count(dl, None, dr, None)
        ok = j == ej and np.array_equal(k.cpu().numpy(), ek) and np.array_equal(c.cpu().numpy(), ec) and \
            np.array_equal(f.cpu().numpy().view(np.uint32).astype(np.int64), ef)
        if not ok:
            bad += 1
            print("MISMATCH", seed, shape, n_l, n_r, off, round_, dev.last_join_form(), hex(int(dev.lib.mdb_dev_last_join_filter(dev.h))), flush=True)
    # single-table GROUP BY of the right table's column (duplicates: the band sort up to 2^25 key values, the tile sort at 2^26 / 2^27, the
    # partitioned path beyond), checked on the device: first rows ascending, COUNT and first row of every key
    for round_ in range(2):
        gf, gc = dev.group_count(dr, None)
        fi = gf.to(torch.int64) & 0xFFFFFFFF
        keys = dr[fi]
        uk, uc = torch.unique(dr, return_counts=True)
        o = torch.argsort(keys)
        first = torch.full((uk.numel(),), n_r, dtype=torch.int64, device=dr.device)
        first.scatter_reduce_(0, torch.searchsorted(uk, dr), torch.arange(n_r, device=dr.device), "amin")
        ok = bool((fi[1:] > fi[:-1]).all()) and keys.numel() == uk.numel() and torch.equal(keys[o], uk) and torch.equal(gc[o], uc) and torch.equal(first, fi[o])
        if not ok:
            bad += 1
            print("GROUP BY MISMATCH", seed, shape, n_r, off, round_, dev.last_plan(), flush=True)
        del gf, gc, fi, keys, uk, uc, o, first
    print(f"case {seed}: shape {shape}, {n_l} x {n_r} rows, {len(ek)} groups, form {dev.last_join_form()}, flags {hex(int(dev.lib.mdb_dev_last_join_filter(dev.h)))}", flush=True)
    del dl, dr, k, c, f
    torch.cuda.empty_cache()
print(f"{cases} cases, {bad} mismatches, {time.time() - t0:.1f} s")
