"""Randomised soak of GROUP BY / SELECT DISTINCT over several columns (mdb_dev_group_count_multi, mdb_dev_distinct_sel) against numpy:
1 ... 5 columns of 1 ... 2^30 values each (composites of a few bits - the LDS tables -, of 18 ... 25 - the band sort that reads the columns
itself -, of up to 63 - a composite column - and beyond - the sort of the stream), INT64 and DOUBLE, NULL bitmaps, DESC flags, a row-id
vector, a hot combination, a value the range sample does not see; device buffers refilled in place between cases; the sample of the ranges
switched on from 2^18 rows (MDB_SORT_RANGE_SAMPLE=2) in every second case."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from midoridb_amd import dev as D
dev = D.DeviceCtx(0)
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sizes = [262_144, 700_001, 2_300_001, 4_194_304, 5_500_000]
bufs = {(n, c): torch.empty(n, dtype=torch.int64, device=dev.device) for n in sizes for c in range(5)}


def numpy_groups(cols, rid, n):
    parts = []
    for v, nl in cols:
        bits = np.ascontiguousarray(v).view(np.uint64)
        if rid is not None:
            bits = bits[rid]
        bits = bits.copy()
        if nl is not None:
            z = nl[rid] if rid is not None else nl
            bits[z] = 0
            parts.append(z.astype(np.uint64))
        parts.append(bits)
    _, first, cnt = np.unique(np.stack(parts, axis=1), axis=0, return_index=True, return_counts=True)
    order = np.argsort(first, kind="stable")
    return first[order].astype(np.int64), cnt[order].astype(np.int64)


bad = 0
forms = {}
t0 = time.time()
for seed in range(seeds):
    rng = np.random.default_rng(70_000 + seed)
    os.environ["MDB_SORT_RANGE_SAMPLE"] = "2" if seed % 2 else "1"
    n = int(rng.choice(sizes))
    nk = int(rng.choice([1, 2, 2, 2, 3, 3, 4, 5]))
    budget = float(rng.choice([10, 14, 17, 22, 25, 40, 63, 70]))        # bits of the composite, roughly
    cols, keys_dev = [], []
    for c in range(nk):
        bits = max(1.0, budget / nk + rng.uniform(-2, 2))
        card = int(min(2**30, max(2, 2 ** bits)))
        is_double = rng.random() < 0.15
        if is_double:
            v = (1.0 + rng.integers(0, min(card, 2**20), n) * 2.0**-52) * float(rng.choice([1.0, -1.0]))
        else:
            v = rng.integers(0, card, n, dtype=np.int64) + int(rng.choice([0, -card // 2, 10**12, -(2**62)]))
        nl = (rng.random(n) < float(rng.choice([0.005, 0.2, 1.0]))) if rng.random() < 0.3 else None
        cols.append((v, nl))
    shape = rng.choice(["plain", "plain", "hot", "outlier", "rid"])
    if shape == "hot":
        hot = rng.random(n) < 0.4
        for v, _ in cols:
            v[hot] = v[0]
    elif shape == "outlier" and not np.issubdtype(cols[-1][0].dtype, np.floating):
        cols[-1][0][int(rng.integers(0, n))] += int(rng.choice([-10**6, 10**6]))
    rid = rng.integers(0, n, n).astype(np.uint32) if shape == "rid" else None
    ridd = dev.to_dev(rid) if rid is not None else None
    for c, (v, nl) in enumerate(cols):
        b = bufs[(n, c)]
        b.copy_(torch.from_numpy(np.ascontiguousarray(v).view(np.int64)))
        keys_dev.append((b.view(torch.float64) if v.dtype == np.float64 else b, dev.nullbits_dev(nl) if nl is not None else None, ridd,
                         D.T_DOUBLE if v.dtype == np.float64 else D.T_INT64, bool(rng.random() < 0.3)))
    dev.prof_enable(True)
    dev.prof_reset()
    first, cnt = dev.group_count_multi(keys_dev, n)
    prof = dev.prof_read()
    dev.prof_enable(False)
    ran = lambda name: name in prof and prof[name][1] > 0
    form = "+".join(f for f, k in (("columns->band sort", "group_band_sort_columns"), ("columns->LDS tables", "group_direct_columns"),
                                   ("composite column", "groupby_pack"), ("sort of the stream", "groupby_heads")) if ran(k))
    forms[form] = forms.get(form, 0) + 1
    ef, ec = numpy_groups(cols, rid.astype(np.int64) if rid is not None else None, n)
    ok = np.array_equal(first.cpu().numpy().view(np.uint32).astype(np.int64), ef) and np.array_equal(cnt.cpu().numpy(), ec)
    sel = dev.distinct_sel(keys_dev, n)
    ok = ok and np.array_equal(sel.cpu().numpy().view(np.uint32).astype(np.int64), ef)
    bad += not ok
    if not ok:
        print("MISMATCH seed", seed, "n", n, "columns", nk, "budget", budget, shape, form)
print(f"{seeds} cases, {bad} mismatches, {time.time() - t0:.1f} s, forms: {forms}")
sys.exit(1 if bad else 0)
