"""Randomised soak of mdb_dev_join_payload against numpy, both forms (MDB_ROWJOIN=2: the row-order form on tables of any size; 0: the older
forms): key windows of 2^15 ... 2^27 values anywhere in the int64 range, unique and foreign-key left sides, one and two payload columns
(INT64 and DOUBLE bit patterns), hot keys, left rows without partner and duplicate right keys (-> "not served", never a wrong answer),
ragged sizes; the same device buffers refilled in place between cases (stale memos, stale statistics-free windows).
    python tests/soak/payload_soak.py [cases]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from midoridb_amd import dev as D
dev = D.DeviceCtx(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = served = stale = 0
t0 = time.time()
for seed in ([int(x) for x in os.environ["SOAK_SEEDS"].split(",")] if os.environ.get("SOAK_SEEDS") else range(cases)):
    rng = np.random.default_rng(70_000 + seed)
    bits = int(rng.integers(15, 28))
    span = 1 << bits
    n_r = int(min(span, rng.choice([40_000, 400_000, 1_500_000, 2_500_000])))
    base = int(rng.choice([0, -5 * 10**9, 10**15]))
    kr = (rng.choice(span, n_r, replace=False) if span <= 4_000_000 else np.unique(rng.integers(0, span, n_r))).astype(np.int64) + base
    n_l = int(rng.choice([1_048_576 + 7, 32768 * 40, 2_345_679, 3_100_001]))
    kl = kr[rng.integers(0, len(kr), n_l)] if rng.random() < 0.6 else np.resize(rng.permutation(kr), n_l)
    kind = rng.choice(["ok", "ok", "ok", "hot", "no_partner", "dup_right"])
    expect = True
    if kind == "hot":
        a = int(rng.integers(0, n_l - 200_000))
        kl[a:a + 150_000] = kr[int(rng.integers(0, len(kr)))]
    elif kind == "no_partner":
        kl[int(rng.integers(0, n_l))] = base - 12345
        expect = False
    elif kind == "dup_right":
        kr = kr.copy()
        kr[1] = kr[0]
        expect = False
    cells = int(rng.integers(1, 3))
    pay = [rng.integers(-2**62, 2**62, len(kr), dtype=np.int64), rng.standard_normal(len(kr))][:cells]
    order = np.argsort(kr, kind="stable")
    pos = np.searchsorted(kr[order], kl)
    pos = np.minimum(pos, len(kr) - 1)
    for form in ("2", "0"):
        os.environ["MDB_ROWJOIN"] = form
        if os.environ.get("SOAK_VERBOSE"):
            print("case", seed, "form", form, "bits", bits, "n_l", n_l, "n_r", len(kr), kind, "cells", cells, flush=True)
        held = (dev.to_dev(kl), dev.to_dev(kr), [dev.to_dev(p) for p in pay])
        got = dev.join_payload(held[0], None, held[1], None, held[2])
        if got is None and expect and kind != "hot":
            # what the operator remembers is keyed by ADDRESS and length, and this loop refills the same addresses: a remembered "these columns
            # are no such join" from an earlier case costs a "not served" (never a wrong answer).  At addresses it has not seen it must serve.
            stale += 1
            fresh = lambda a: dev.to_dev(np.concatenate([a[:2 * (1 + seed % 5)], a]))[2 * (1 + seed % 5):]     # (inside a block, where no column began)
            got = dev.join_payload(fresh(kl), None, fresh(kr), None, [fresh(p) for p in pay])
        if not expect:
            ok = got is None
        elif got is None:
            ok = kind == "hot"        # (fixed-capacity regions may overflow under a hot key - the older forms, and windows past 2^27 values: not served, by design)
        else:
            served += 1
            ok = all(np.array_equal(g.cpu().numpy().view(np.int64), p[order][pos].view(np.int64)) for g, p in zip(got, pay))
        bad += not ok
        if not ok:
            print("MISMATCH seed", seed, "form", form, "bits", bits, "n_l", n_l, "n_r", len(kr), kind, "cells", cells,
                  "not served" if got is None else "WRONG VALUES", flush=True)
print(f"{cases} cases x 2 forms, {served} served, {stale} not served on a remembered verdict about other data, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
