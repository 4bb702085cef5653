"""Randomised soak of the fused join + GROUP BY operator (one call, split form, int32 entry points, materialising join)
against the C hash-join oracle and the numpy oracle: random sizes, key windows anywhere in the int64 range, spans below
and above 2^32, duplicate factors, NULL rates, hidden outliers, narrow-key modes 0 / 1 / 2."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from oracle import cpu, np_oracle as orc
from midoridb_amd import dev as D
dev = D.DeviceCtx(0)
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
t0 = time.time()
for seed in range(seeds):
    rng = np.random.default_rng(70_000 + seed)
    mode = int(rng.choice([0, 1, 1, 2]))
    dev.set_narrow_keys(mode)
    n_l = int(10 ** rng.uniform(4.5, 6.7))
    n_r = int(10 ** rng.uniform(4.5, 6.7))
    span = int(2 ** rng.uniform(8, 34))
    off = int(rng.choice([0, 0, -span // 2, 10**12, -(2**60), 2**62 - span - 5]))
    kl = off + rng.integers(0, span, n_l, dtype=np.int64)
    kr = off + rng.integers(0, span, n_r, dtype=np.int64)
    if rng.random() < 0.3:
        kl = off + rng.permutation(max(span, n_l))[:n_l].astype(np.int64) if span >= n_l and span < 3 * 10**7 else kl
    if rng.random() < 0.3:                                   # hidden outliers
        kl[int(rng.integers(0, n_l))] = off + span + 2**33
        kr[int(rng.integers(0, n_r))] = off - 2**35
    nl = (rng.random(n_l) < 0.03) if rng.random() < 0.5 else None
    nr = (rng.random(n_r) < 0.03) if rng.random() < 0.3 else None
    ek, ec, ef, ej = cpu.hash_join_group_count(kl, nl, kr, nr, 8)
    dl, dnl, dr, dnr = dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr), dev.nullbits_dev(nr)
    how = int(rng.integers(0, 3))
    i32ok = nl is None and nr is None and kl.min() >= -2**31 and kl.max() < 2**31 and kr.min() >= -2**31 and kr.max() < 2**31
    if how == 0:
        k, c, f, j = dev.join_group_count(dl, dnl, dr, dnr)
    elif how == 1 or not i32ok:
        dev.join_group_count_begin(dl, dnl, n_r + int(rng.integers(0, 1000)) if rng.random() < 0.8 else n_r // 2)
        k, c, f, j = dev.join_group_count_finish(dr, dnr)
    else:
        k, c, f, j = dev.join_group_count_i32(torch.from_numpy(kl.astype(np.int32)).to(dev.device), torch.from_numpy(kr.astype(np.int32)).to(dev.device))
    ok = (j == ej and k.numel() == len(ek) and np.array_equal(k.cpu().numpy(), ek) and np.array_equal(c.cpu().numpy(), ec)
          and np.array_equal(f.cpu().numpy().view(np.uint32).astype(np.int64), ef))
    # plain GROUP BY of the left column
    first, cnt = dev.group_count(dl, dnl)
    e_first, e_cnt = orc.group_count(kl, nl)
    ok = ok and np.array_equal(first.cpu().numpy().view(np.uint32).astype(np.int64), e_first) and np.array_equal(cnt.cpu().numpy(), e_cnt)
    if seed % 4 == 0 and n_l * 1.0 * n_r / max(span, 1) < 3e7:      # materialising join while the pair count stays small
        el, er = orc.join_pairs(kl, nl, kr, nr)
        l, r = dev.join_pairs(dl, dnl, dr, dnr)
        ok = ok and np.array_equal(l.cpu().numpy().astype(np.int64), el) and np.array_equal(r.cpu().numpy().astype(np.int64), er)
    bad += not ok
    if not ok:
        print("MISMATCH seed", seed, "mode", mode, "n", n_l, n_r, "span", span, "off", off, "how", how)
dev.set_narrow_keys(1)
print(f"{seeds} cases, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
