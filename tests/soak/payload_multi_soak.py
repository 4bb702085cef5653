"""Randomised soak of mdb_dev_join_payload_multi against numpy (round 6): 1 ... 4 right tables on one key with 1 or 2 payload columns each (four
columns at most), key windows of 2^15 ... 2^27 values anywhere in the int64 range, unique and foreign-key left sides, hot keys, and the three ways a
statement about the tables can be false - a left row without partner in ONE of the tables, a right key twice in one of them, a key outside the bound
the caller hands over (-> "not served", never a wrong answer); both grids of the leaf (MDB_RJ_PERSIST).
    python tests/soak/payload_multi_soak.py [cases]"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from midoridb_amd import dev as D
dev = D.DeviceCtx(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
os.environ["MDB_ROWJOIN"] = "2"
bad = served = 0
t0 = time.time()
for seed in range(cases):
    rng = np.random.default_rng(880_000 + seed)
    os.environ["MDB_RJ_PERSIST"] = "1" if seed % 3 else "0"
    bits = int(rng.integers(15, 28))
    span = 1 << bits
    base = int(rng.choice([0, -7 * 10**9, 10**16]))
    n_u = int(min(span, rng.choice([30_000, 300_000, 1_200_000, 2_200_000])))
    universe = (rng.choice(span, n_u, replace=False) if span <= 4_000_000 else np.unique(rng.integers(0, span, n_u))).astype(np.int64) + base
    ntab = int(rng.integers(1, 5))
    cells = [int(rng.integers(1, 3)) for _ in range(ntab)]
    while sum(cells) > 4:
        cells[int(np.argmax(cells))] -= 1
    rights = []
    for t in range(ntab):
        kr = rng.permutation(universe)
        rights.append([kr, [rng.integers(-2**63, 2**63 - 1, len(kr), dtype=np.int64), rng.standard_normal(len(kr))][:cells[t]]])
    n_l = int(rng.choice([40_000, 1_048_576 + 7, 32768 * 30, 2_345_679]))
    kl = universe[rng.integers(0, len(universe), n_l)] if rng.random() < 0.6 else np.resize(rng.permutation(universe), n_l)
    kind = rng.choice(["ok", "ok", "ok", "hot", "no_partner", "dup_right", "outside"])
    lo, hi = base, base + span - 1
    expect = True
    if kind == "hot":
        a = int(rng.integers(0, max(1, n_l - 200_000)))
        kl[a:a + 150_000] = universe[int(rng.integers(0, len(universe)))]
    elif kind == "no_partner":      # the key leaves ONE right table (replaced by a value no table holds, inside the window when there is room)
        t = int(rng.integers(0, ntab))
        victim = kl[int(rng.integers(0, n_l))]
        kr = rights[t][0].copy()
        spare = np.setdiff1d(np.arange(base, base + min(span, len(universe) + 64), dtype=np.int64), universe)
        if len(spare) == 0:
            kind = "ok"
        else:
            kr[kr == victim] = spare[0]
            rights[t][0] = kr
            expect = False
    elif kind == "dup_right":
        t = int(rng.integers(0, ntab))
        kr = rights[t][0].copy()
        gone = kr[1]
        kr[1] = kr[0]
        rights[t][0] = kr
        kl = kl[kl != gone] if (kl != gone).any() else kl
        expect = False
    elif kind == "outside":
        kl = kl.copy()
        kl[int(rng.integers(0, len(kl)))] = hi + 1 + int(rng.integers(0, 1000))
        expect = False
    got = dev.join_payload_multi(dev.to_dev(kl), [(dev.to_dev(kr), [dev.to_dev(p) for p in pay]) for kr, pay in rights], lo, hi)
    if not expect:
        ok = got is None
    elif got is None:
        ok = False
    else:
        served += 1
        ok = True
        for (kr, pay), outs in zip(rights, got):
            order = np.argsort(kr, kind="stable")
            pos = order[np.minimum(np.searchsorted(kr[order], kl), len(kr) - 1)]
            ok = ok and all(np.array_equal(g.cpu().numpy().view(np.int64), p[pos].view(np.int64)) for g, p in zip(outs, pay))
    bad += not ok
    if not ok:
        print("MISMATCH seed", seed, "bits", bits, "tables", ntab, "cells", cells, "n_l", len(kl), kind, "not served" if got is None else "WRONG VALUES", flush=True)
print(f"{cases} cases, {served} served, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
