"""GPU parity tests of the device operators (include/mdb_dev.h) against the numpy oracle.

Every call goes through the C-ABI of libmidoridb_amd.so; results must be bit-exact
(integer / index work).  Sizes are chosen to cover: a single tile, one partition level,
two partition levels (> 393 216 build rows), ragged tails, NULLs, duplicates (N:M), keys
that are negative / zero (the key whose hash is 0 has a dedicated slot), and skew.
"""
import os
import numpy as np
import pytest
import torch

from oracle import np_oracle as orc
from midoridb_amd import dev as D

pytestmark = pytest.mark.gpu


def _mk(rng, n, domain, null_frac=0.0, lo=0):
    keys = rng.integers(lo, lo + max(domain, 1), size=n, dtype=np.int64)
    nulls = None
    if null_frac > 0:
        nulls = rng.random(n) < null_frac
    return keys, nulls


def _np(t):
    return t.cpu().numpy()


CASES_JGC = [
    # (n_l, n_r, domain, null_frac, lo)
    (1, 1, 1, 0.0, 0),
    (5, 7, 3, 0.0, 0),
    (63, 65, 10, 0.2, -5),
    (64, 64, 64, 0.0, 0),
    (1000, 1500, 200, 0.1, -100),
    (4096, 4097, 5000, 0.05, 0),
    (50_000, 70_000, 20_000, 0.01, -10_000),
    (300_000, 300_000, 300_000, 0.0, 0),
    (1_000_000, 1_200_000, 700_000, 0.02, -350_000),     # two partition levels
    (2_000_000, 500_000, 3_000_000, 0.0, 0),
]


@pytest.mark.parametrize("n_l,n_r,domain,null_frac,lo", CASES_JGC)
def test_join_group_count(dev, n_l, n_r, domain, null_frac, lo):
    rng = np.random.default_rng(n_l * 31 + n_r)
    kl, nl = _mk(rng, n_l, domain, null_frac, lo)
    kr, nr = _mk(rng, n_r, domain, null_frac, lo)
    ek, ec, ef, ej = orc.join_group_count(kl, nl, kr, nr)
    k, c, f, j = dev.join_group_count(dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr), dev.nullbits_dev(nr))
    assert j == ej
    assert np.array_equal(_np(f).astype(np.int64), ef)
    assert np.array_equal(_np(k), ek)
    assert np.array_equal(_np(c), ec)


def test_join_group_count_golden_case11(dev):
    # reference tests/engine/executor_select.c:348-378 : A(id_a)=1,3,4 ; B(id_b)=1,1,3,3,4,NULL
    kl = np.array([1, 3, 4], dtype=np.int64)
    kr = np.array([1, 1, 3, 3, 4, 0], dtype=np.int64)
    nr = np.array([0, 0, 0, 0, 0, 1], dtype=bool)
    k, c, f, j = dev.join_group_count(dev.to_dev(kl), None, dev.to_dev(kr), dev.nullbits_dev(nr))
    assert _np(k).tolist() == [1, 3, 4]
    assert _np(c).tolist() == [2, 2, 1]
    assert j == 5


def test_join_group_count_empty_and_disjoint(dev):
    z = dev.to_dev(np.zeros(0, dtype=np.int64))
    a = dev.to_dev(np.arange(10, dtype=np.int64))
    b = dev.to_dev(np.arange(100, 110, dtype=np.int64))
    for l, r in ((z, a), (a, z), (a, b)):
        k, c, f, j = dev.join_group_count(l, None, r, None)
        assert k.numel() == 0 and j == 0


def test_join_group_count_skew(dev):
    # one hot key on both sides (N:M = 3000 x 5000) plus the zero key and int64 extremes
    rng = np.random.default_rng(7)
    kl = np.concatenate([np.full(3000, 42), np.zeros(10), rng.integers(-2**62, 2**62, 20_000),
                         [np.iinfo(np.int64).min, np.iinfo(np.int64).max]]).astype(np.int64)
    kr = np.concatenate([np.full(5000, 42), np.zeros(3), kl[3010:13010],
                         [np.iinfo(np.int64).min]]).astype(np.int64)
    rng.shuffle(kl)
    rng.shuffle(kr)
    ek, ec, ef, ej = orc.join_group_count(kl, None, kr, None)
    k, c, f, j = dev.join_group_count(dev.to_dev(kl), None, dev.to_dev(kr), None)
    assert j == ej
    assert np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec) and np.array_equal(_np(f).astype(np.int64), ef)


def test_join_group_count_skew_two_levels_falls_back_to_exact_layout(dev):
    """A key repeated 300 000 times in a 10^6-row table overflows its fixed-capacity leaf region of the
    histogram-free layout; the operator must notice and redo the partitioning with exact histograms."""
    rng = np.random.default_rng(21)
    kl = np.concatenate([np.full(300_000, 7), rng.integers(0, 10**9, 700_000)]).astype(np.int64)
    kr = np.concatenate([np.full(200_000, 7), rng.integers(0, 10**9, 600_000), kl[300_000:500_000]]).astype(np.int64)
    rng.shuffle(kl)
    rng.shuffle(kr)
    ek, ec, ef, ej = orc.join_group_count(kl, None, kr, None)
    k, c, f, j = dev.join_group_count(dev.to_dev(kl), None, dev.to_dev(kr), None)
    assert j == ej
    assert np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec) and np.array_equal(_np(f).astype(np.int64), ef)
    g_f, g_c = dev.group_count(dev.to_dev(kl), None)
    e_f, e_c = orc.group_count(kl, None)
    assert np.array_equal(_np(g_f).astype(np.int64), e_f) and np.array_equal(_np(g_c), e_c)


@pytest.mark.parametrize("n,domain,null_frac", [(1, 1, 0.0), (10, 3, 0.5), (65, 7, 0.3), (5000, 50, 0.1),
                                                  (100_000, 100_000, 0.0), (1_500_000, 400_000, 0.05)])
def test_group_count(dev, n, domain, null_frac):
    rng = np.random.default_rng(n)
    k, nl = _mk(rng, n, domain, null_frac, -domain // 2)
    ef, ec = orc.group_count(k, nl)
    f, c = dev.group_count(dev.to_dev(k), dev.nullbits_dev(nl))
    assert np.array_equal(_np(f).astype(np.int64), ef)
    assert np.array_equal(_np(c), ec)


@pytest.mark.parametrize("case", ["uniform_first_rows_2e22", "bunched_first_rows_2e20", "unique_2e23", "sparse_2e26", "hot_value", "far_from_zero"])
def test_group_count_through_the_tile_sort(dev, case, monkeypatch):
    """GROUP BY key + COUNT(*) of a NULL-free column through the tile sort (mdb_group_count_tiled, MDB_GROUP_TILED=1; off by default -
    measured no faster than the partitioned path): first row and COUNT per key in first-row order, bit-exact against the oracle, for both
    ways the groups leave the leaf kernel - the ordering kernel's row-id ranges (first rows spread evenly) and the record list + sort"""
    rng = np.random.default_rng(len(case))
    n = 2_300_000
    if case == "uniform_first_rows_2e22":      # most keys once or twice: first rows everywhere -> ranged emit
        k = rng.integers(0, 1 << 22, n)
    elif case == "bunched_first_rows_2e20":    # ~2 rows per key on average over 2^20 values, every key's first row early
        k = np.concatenate([rng.permutation(1 << 20), rng.integers(0, 1 << 20, n - (1 << 20))])
    elif case == "unique_2e23":
        k = rng.permutation(1 << 23)[:n]
    elif case == "sparse_2e26":
        k = rng.integers(0, 1 << 26, n)
    elif case == "hot_value":                  # 300 000 consecutive rows of one value: pieces of whole tiles in one digit
        k = rng.integers(0, 1 << 21, n)
        k[900_000:1_200_000] = 777
    else:
        k = rng.integers(0, 1 << 21, n) + 10**13
    k = k.astype(np.int64)
    ef, ec = orc.group_count(k, None)
    monkeypatch.setenv("MDB_GROUP_TILED", "1")
    monkeypatch.setenv("MDB_GROUP_BANDED", "0")     # (the band sort would serve most of these windows first)
    dev.prof_enable(True)
    dev.prof_reset()
    f, c = dev.group_count(dev.to_dev(k), None)
    ran = set(dev.prof_read())
    dev.prof_enable(False)
    assert {"group_tile_sort", "group_tile_leaf"} <= ran, ran
    assert ("order_leaf_sparse" in ran) == (case in ("uniform_first_rows_2e22", "hot_value", "far_from_zero")) or "order_leaf" in ran, ran
    assert np.array_equal(_np(f).astype(np.int64), ef) and np.array_equal(_np(c), ec), case


@pytest.mark.parametrize("case", ["groups_of_17_2e18", "uniform_2e21", "bunched_first_rows_2e22", "sparse_2e24", "whole_bands", "hot_value", "negative_keys"])
def test_group_count_through_the_band_sort(dev, case, monkeypatch):
    """GROUP BY key + COUNT(*) of a NULL-free column through the band sort (mdb_dev_bandgroup.hip: 4-byte row words, the default for windows of
    2^18 ... 2^25 key values from 2^21 rows on): first row and COUNT per key in first-row order, bit-exact against the oracle - a ragged last tile
    and band, windows at both ends of the range, first rows bunched at the table's start, keys far from zero; a hot value overflows its region:
    the other forms answer (same result), and the column is remembered"""
    rng = np.random.default_rng(len(case) + 5)
    n = 4_500_000 + 777
    if case == "groups_of_17_2e18":
        k = rng.integers(0, 1 << 18, n)
    elif case == "uniform_2e21":
        k = rng.integers(0, 1 << 21, n)
    elif case == "bunched_first_rows_2e22":
        k = np.concatenate([rng.permutation(1 << 22), rng.integers(0, 1 << 22, n - (1 << 22))])
    elif case == "sparse_2e24":                # (a sampled window is padded: 2^25 values, the form's widest)
        k = rng.integers(0, 1 << 24, n)
    elif case == "whole_bands":
        n = 18 << 18
        k = rng.integers(0, 3_000_000, n)
    elif case == "hot_value":                  # 400 000 rows of one value within two bands: its digit's regions there overflow
        n += 4096                              # (what is remembered about a column goes by address and length: not the next case's column)
        k = rng.integers(0, 1 << 21, n)
        k[900_000:1_300_000] = 777
    else:
        k = rng.integers(0, 1 << 20, n) - 10**13
    k = k.astype(np.int64)
    ef, ec = orc.group_count(k, None)
    dk = dev.to_dev(k)
    dev.prof_enable(True)
    dev.prof_reset()
    f, c = dev.group_count(dk, None)
    ran = {k for k, v in dev.prof_read().items() if v[0] > 0}
    dev.prof_reset()
    f2, c2 = dev.group_count(dk, None)
    ran2 = {k for k, v in dev.prof_read().items() if v[0] > 0}
    dev.prof_enable(False)
    assert {"group_band_sort", "group_band_leaf"} <= ran, ran
    assert any(k.startswith(("leaf_", "hot_")) for k in ran) == (case == "hot_value"), ran       # (the partitioned path's kernels)
    assert dev.last_plan()["group_form"] == (0 if case == "hot_value" else 1), dev.last_plan()    # (of the second call)
    assert ("group_band_sort" in ran2) == (case != "hot_value"), ran2
    assert np.array_equal(_np(f).astype(np.int64), ef) and np.array_equal(_np(c), ec), case
    assert np.array_equal(_np(f2).astype(np.int64), ef) and np.array_equal(_np(c2), ec), case
    monkeypatch.setenv("MDB_GROUP_BANDED", "0")
    f3, c3 = dev.group_count(dk, None)
    assert torch.equal(f3, f) and torch.equal(c3, c)


@pytest.mark.parametrize("span_bits", [26, 27])
def test_group_count_windows_of_2e26_and_2e27_values_go_through_the_tile_sort(dev, span_bits):
    """Where the band sort does not reach (windows of 2^26 and 2^27 key values) large tables take the tile sort by default
    (mdb_group_count_tiled): 2 x 10^7 rows, ~1.3 rows per key, checked on the device against torch - COUNT per key, first row per key,
    first-occurrence order"""
    n, span = 20_000_000 + 4097, 1 << span_bits
    g = torch.Generator(device="cuda")
    g.manual_seed(span_bits)
    kl = torch.randint(0, span, (n,), dtype=torch.int64, device="cuda", generator=g) - 5_000_000_000
    dev.prof_enable(True)
    dev.prof_reset()
    dev.call_stats(kl, (-5_000_000_000, -5_000_000_000 + span - 1))     # (the catalog's range, as query_execute() hands it over: a sampled window is padded
    try:                                                                  #  and may come out as 2^28 values, which no 4-byte-word form serves)
        f, c = dev.group_count(kl, None)
    finally:
        dev.call_stats()
    ran = {k for k, v in dev.prof_read().items() if v[0] > 0}
    dev.prof_enable(False)
    assert {"group_tile_sort", "group_tile_leaf"} <= ran and "group_band_sort" not in ran, ran
    fi = f.to(torch.int64) & 0xFFFFFFFF
    assert bool((fi[1:] > fi[:-1]).all())
    keys = kl[fi]
    uk, uc = torch.unique(kl, return_counts=True)
    o = torch.argsort(keys)
    assert torch.equal(keys[o], uk) and torch.equal(c[o], uc)
    first = torch.full((uk.numel(),), n, dtype=torch.int64, device="cuda")
    first.scatter_reduce_(0, torch.searchsorted(uk, kl), torch.arange(n, device="cuda"), "amin")
    assert torch.equal(first, fi[o])


@pytest.mark.parametrize("form", ["tile_sort", "band_sort"])
@pytest.mark.parametrize("case", ["unique", "one_percent_pairs", "some_triples_and_a_run", "eight_percent_duplicates"])
def test_group_count_nearly_unique_keys_leave_as_bits_and_exceptions(dev, case, form, monkeypatch):
    """The tile-sorted GROUP BY over a column whose keys are nearly unique (a pilot over 64 key digits counts the rows that are not the
    first of their key: one in 16 at most): no record per group and no sort of records - one bit per row, cleared for every row that is
    not its key's first, plus (first row, COUNT) exceptions for the keys with several rows, expanded by mdb_dev_dense.hip.  Bit-exact
    against the oracle; more duplicates than that: the record form, same result."""
    rng = np.random.default_rng(len(case) + 31)
    n = 4_400_000 + 123 + (4096 if form == "band_sort" else 0)
    k = rng.permutation(1 << (26 if form == "tile_sort" else 23))[:n].astype(np.int64) + 7_000_000_000
    if case == "one_percent_pairs":
        src = rng.integers(0, n, n // 100)
        k[rng.integers(0, n, n // 100)] = k[src]
    elif case == "some_triples_and_a_run":
        for _ in range(2):
            src = rng.integers(0, n, 20_000)
            k[rng.integers(0, n, 20_000)] = k[src]
        run = 3000 if form == "tile_sort" else 100     # (hundreds of rows of one key inside one band overflow its region: the band sort's other test)
        k[1_000_000:1_000_000 + run] = k[17]           # a run of rows of one key: one exception with a large COUNT
    elif case == "eight_percent_duplicates":
        src = rng.integers(0, n, n // 12)
        k[rng.integers(0, n, n // 12)] = k[src]
    ef, ec = orc.group_count(k, None)
    if form == "tile_sort":
        monkeypatch.setenv("MDB_GROUP_TILED", "1")
        monkeypatch.setenv("MDB_GROUP_BANDED", "0")
    leaf = "group_tile_leaf" if form == "tile_sort" else "group_band_leaf"
    dk = dev.to_dev(k)
    dev.prof_enable(True)
    dev.prof_reset()
    f, c = dev.group_count(dk, None)
    ran = {kk for kk, v in dev.prof_read().items() if v[0] > 0}
    dev.prof_enable(False)
    assert leaf + "_dense" in ran, ran                                # (the pilot at least)
    plan = dev.last_plan()
    assert plan["group_form"] == (2 if form == "tile_sort" else 1) and plan["groups_as_bits"] == (0 if case == "eight_percent_duplicates" else 1), plan
    assert ("dense_expand" in ran) == (case != "eight_percent_duplicates"), ran
    assert (leaf in ran) == (case == "eight_percent_duplicates"), ran
    assert np.array_equal(_np(f).astype(np.int64), ef) and np.array_equal(_np(c), ec), case
    monkeypatch.setenv("MDB_GROUP_DENSE", "0")
    f2, c2 = dev.group_count(dk, None)
    assert torch.equal(f2, f) and torch.equal(c2, c)


def test_group_count_golden_case10(dev):
    # reference tests/engine/executor_select.c:318-346 : id = 1,1,3,3,4 -> (1,2)(3,2)(4,1)
    k = np.array([1, 1, 3, 3, 4], dtype=np.int64)
    f, c = dev.group_count(dev.to_dev(k), None)
    assert _np(f).tolist() == [0, 2, 4] and _np(c).tolist() == [2, 2, 1]


def test_group_count_all_null(dev):
    k = np.zeros(100, dtype=np.int64)
    nl = np.ones(100, dtype=bool)
    f, c = dev.group_count(dev.to_dev(k), dev.nullbits_dev(nl))
    assert _np(f).tolist() == [0] and _np(c).tolist() == [100]


CASES_PAIRS = [
    (1, 1, 1, 0.0), (3, 2, 3, 0.0), (65, 63, 9, 0.2), (2000, 3000, 500, 0.1), (40_000, 60_000, 30_000, 0.02),
    (700_000, 900_000, 1_000_000, 0.0),   # two partition levels (900k right rows / 640 per leaf)
]


@pytest.mark.parametrize("n_l,n_r,domain,null_frac", CASES_PAIRS)
def test_join_pairs(dev, n_l, n_r, domain, null_frac):
    rng = np.random.default_rng(n_l + 3 * n_r)
    kl, nl = _mk(rng, n_l, domain, null_frac, -3)
    kr, nr = _mk(rng, n_r, domain, null_frac, -3)
    el, er = orc.join_pairs(kl, nl, kr, nr)
    l, r = dev.join_pairs(dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr), dev.nullbits_dev(nr))
    assert l.numel() == len(el)
    assert np.array_equal(_np(l).astype(np.int64), el)
    assert np.array_equal(_np(r).astype(np.int64), er)


def test_join_pairs_golden_case3(dev):
    # reference tests/engine/executor_select.c:102-131 : A.id_a=1,2,3 ; B.id_b=1,3 -> rows (0,0),(2,1)
    l, r = dev.join_pairs(dev.to_dev(np.array([1, 2, 3], dtype=np.int64)), None,
                          dev.to_dev(np.array([1, 3], dtype=np.int64)), None)
    assert _np(l).tolist() == [0, 2] and _np(r).tolist() == [0, 1]


def test_join_pairs_heavy_duplicates(dev):
    # one key 300 x 5000 times: exercises the chunked emit (right side > 2048 rows of one key)
    kl = np.concatenate([np.full(300, 9), np.arange(100, 1100)]).astype(np.int64)
    kr = np.concatenate([np.full(5000, 9), np.arange(100, 600), np.zeros(70)]).astype(np.int64)
    rng = np.random.default_rng(3)
    rng.shuffle(kl)
    rng.shuffle(kr)
    el, er = orc.join_pairs(kl, None, kr, None)
    l, r = dev.join_pairs(dev.to_dev(kl), None, dev.to_dev(kr), None)
    assert np.array_equal(_np(l).astype(np.int64), el) and np.array_equal(_np(r).astype(np.int64), er)


def test_cross_pairs(dev):
    l, r = dev.cross_pairs(3, 70)
    assert _np(l).tolist() == [i for i in range(3) for _ in range(70)]
    assert _np(r).tolist() == [j for _ in range(3) for j in range(70)]


def test_filter_programs(dev):
    rng = np.random.default_rng(11)
    n = 100_003
    a = rng.integers(-1000, 1000, n, dtype=np.int64)
    b = rng.integers(-1000, 1000, n, dtype=np.int64)
    x = rng.random(n)
    na = rng.random(n) < 0.1
    nb = rng.random(n) < 0.1
    cols_np = [(a, na, None), (b, nb, None), (x, None, None)]
    cols_dev = [(dev.to_dev(a), dev.nullbits_dev(na), None), (dev.to_dev(b), dev.nullbits_dev(nb), None),
                (dev.to_dev(x), None, None)]
    half = int(np.float64(0.5).view(np.int64))
    progs = [
        [(D.P_CMP_COL_CONST, D.CMP_GT, D.T_INT64, 0, 0, 500)],
        [(D.P_CMP_CONST_COL, D.CMP_GE, D.T_INT64, 0, 0, 123), (D.P_CMP_COL_CONST, D.CMP_LT, D.T_INT64, 0, 0, 200), (D.P_AND, 0, 0, 0, 0, 0)],
        [(D.P_CMP_COL_COL, D.CMP_EQ, D.T_INT64, 0, 1, 0)],
        [(D.P_CMP_COL_COL, D.CMP_NE, D.T_INT64, 0, 1, 0), (D.P_ISNULL, 0, 0, 1, 0, 0), (D.P_OR, 0, 0, 0, 0, 0)],
        [(D.P_ISNULL, 1, 0, 0, 0, 0), (D.P_CMP_COL_CONST, D.CMP_LE, D.T_DOUBLE, 2, 0, half), (D.P_XOR, 0, 0, 0, 0, 0)],
        [(D.P_CONST, 0, 0, 0, 0, 1), (D.P_CMP_COL_CONST, D.CMP_EQ, D.T_INT64, 1, 0, -7), (D.P_AND, 0, 0, 0, 0, 0)],
        [(D.P_CONST, 0, 0, 0, 0, 0)],
    ]
    for prog in progs:
        exp = orc.filter_positions([(op, c, t, aa, bb, (np.float64(0.5) if (t == D.T_DOUBLE) else imm))
                                    for (op, c, t, aa, bb, imm) in prog], cols_np, n)
        got = _np(dev.filter(prog, cols_dev, n)).astype(np.int64)
        assert np.array_equal(got, exp), prog


@pytest.mark.parametrize("n", [262_144, 262_145, 1_000_003, 6_000_011])
def test_filter_one_comparison_single_pass_selection_vector(dev, n):
    """From 2^18 rows on, ONE comparison of an INT64 base-table column with a constant produces its selection vector in one
    pass (k_scan_project_cmp1 with the positions as the only output, decoupled look-back over 16384-row blocks): all six
    operators, with and without a NULL bitmap, every / no / one row passing, ragged tails."""
    rng = np.random.default_rng(n % 977)
    a = rng.integers(-1000, 1000, n, dtype=np.int64)
    na = rng.random(n) < 0.07
    for nulls in (None, na):
        cols_np, cols_dev = [(a, nulls, None)], [(dev.to_dev(a), dev.nullbits_dev(nulls), None)]
        for cmp_, imm in ((D.CMP_GT, 0), (D.CMP_LT, -300), (D.CMP_GE, 999), (D.CMP_LE, -1000), (D.CMP_EQ, 17), (D.CMP_NE, 17), (D.CMP_GT, 5000), (D.CMP_GE, -5000)):
            prog = [(D.P_CMP_COL_CONST, cmp_, D.T_INT64, 0, 0, imm)]
            exp = orc.filter_positions(prog, cols_np, n)
            got = _np(dev.filter(prog, cols_dev, n)).astype(np.int64)
            assert np.array_equal(got, exp), (n, cmp_, imm, nulls is not None)


def test_filter_through_rid_vector(dev):
    rng = np.random.default_rng(5)
    a = rng.integers(0, 50, 1000, dtype=np.int64)
    na = rng.random(1000) < 0.2
    rid = rng.integers(0, 1000, 7777).astype(np.int32)
    prog = [(D.P_CMP_COL_CONST, D.CMP_LT, D.T_INT64, 0, 0, 25)]
    exp = orc.filter_positions(prog, [(a, na, rid)], len(rid))
    got = _np(dev.filter(prog, [(dev.to_dev(a), dev.nullbits_dev(na), dev.to_dev(rid))], len(rid))).astype(np.int64)
    assert np.array_equal(got, exp)


def test_gather64_with_nulls(dev):
    rng = np.random.default_rng(9)
    src = rng.integers(-10**12, 10**12, 5000, dtype=np.int64)
    nul = rng.random(5000) < 0.3
    idx = rng.integers(0, 5000, 12_345).astype(np.int32)
    out, onull = dev.gather64(dev.to_dev(src), dev.nullbits_dev(nul), dev.to_dev(idx), len(idx))
    assert np.array_equal(_np(out), src[idx])
    assert np.array_equal(D.unpack_nullbits(_np(onull).view(np.uint64), len(idx)), nul[idx])
    out2, _ = dev.gather64(dev.to_dev(src), None, None, 5000)
    assert np.array_equal(_np(out2), src)


@pytest.mark.parametrize("nrids", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("n", [1000, 1024, 4096, 70_001])
def test_gather_cols_every_row_id_count_full_and_partial_blocks(dev, nrids, n):
    """mdb_dev_gather_cols has kernel instances that load 1, 2, 4 or 8 row-id vectors per output row; blocks of 1024 outputs
    that all exist issue their loads together, the last (partial) block keeps the per-row range tests.  Every column equals
    the oracle's gather, NULL bits included, with each number of row-id vectors and with full / partial / no full blocks."""
    rng = np.random.default_rng(nrids * 1000 + n)
    m = 50_000
    rids = [rng.integers(0, m, n, dtype=np.int64).astype(np.int32) for _ in range(nrids)]
    d_rids = [dev.to_dev(r) for r in rids]
    cols, want = [], []
    for c in range(nrids + 1):              # one column per row-id vector + one identity column
        src = rng.integers(-2**62, 2**62, max(m, n), dtype=np.int64)
        nulls = rng.random(max(m, n)) < 0.2 if c % 2 == 0 else None
        rid = rids[c] if c < nrids else None
        cols.append((dev.to_dev(src), dev.nullbits_dev(nulls), d_rids[c] if c < nrids else None))
        idx = rid.astype(np.int64) if rid is not None else np.arange(n)
        want.append((src[idx], None if nulls is None else nulls[idx]))
    got = dev.gather_cols(cols, n)
    for c, ((v, nb), (ev, en)) in enumerate(zip(got, want)):
        assert np.array_equal(_np(v), ev), (nrids, n, c)
        assert (nb is None) == (en is None), (nrids, n, c)
        if en is not None:
            assert np.array_equal(D.unpack_nullbits(_np(nb).view(np.uint64), n), en), (nrids, n, c)


def test_gather_cols_whole_projection_in_one_launch(dev):
    """mdb_dev_gather_cols: 16 columns over 3 row-id vectors (and identity) in one launch = 16 separate gather64 calls."""
    rng = np.random.default_rng(12)
    n_src, n = 70_001, 123_457
    rids = [rng.integers(0, n_src, n).astype(np.int32) for _ in range(3)] + [None]
    d_rids = [dev.to_dev(r) if r is not None else None for r in rids]	# columns of one table share ONE row-id vector
    cols, exp = [], []
    for c in range(16):
        src = rng.integers(-10**15, 10**15, n_src if rids[c % 4] is not None else n, dtype=np.int64)
        nul = (rng.random(len(src)) < 0.2) if c % 3 else None
        rid = rids[c % 4]
        cols.append((dev.to_dev(src), dev.nullbits_dev(nul), d_rids[c % 4]))
        idx = rid if rid is not None else np.arange(n)
        exp.append((src[idx], None if nul is None else nul[idx]))
    got = dev.gather_cols(cols, n)
    for (v, nb), (ev, en) in zip(got, exp):
        assert np.array_equal(_np(v), ev)
        assert (nb is None) == (en is None)
        if en is not None:
            assert np.array_equal(D.unpack_nullbits(_np(nb).view(np.uint64), n), en)
    # DOUBLE payload is moved as opaque 8-byte cells
    x = rng.normal(0, 1, 1000)
    (v, _), = dev.gather_cols([(dev.to_dev(x), None, dev.to_dev(np.arange(999, -1, -1, dtype=np.int32)))], 1000)
    assert np.array_equal(_np(v).view(np.int64), x[::-1].view(np.int64))


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4095, 4096, 4097, 300_001])
def test_filter_project_scan_where_projection(dev, n):
    """mdb_dev_filter_project: the predicate bitmap goes straight into the compacted output columns (values and NULL bits in
    row order) - equal to filter() + gather64 per column."""
    rng = np.random.default_rng(n)
    a = rng.integers(-50, 50, n, dtype=np.int64)
    b = rng.normal(0, 1, n)
    na, nb_ = rng.random(n) < 0.1, rng.random(n) < 0.3
    prog = [(D.P_CMP_COL_CONST, D.CMP_GT, D.T_INT64, 0, 0, -10), (D.P_CMP_COL_CONST, D.CMP_LT, D.T_DOUBLE, 1, 0, 0.5), (D.P_AND, 0, 0, 0, 0, 0)]
    da, db, dna, dnb = dev.to_dev(a), dev.to_dev(b), dev.nullbits_dev(na), dev.nullbits_dev(nb_)
    keep = (a > -10) & ~na & (b < 0.5) & ~nb_
    m, outs = dev.filter_project(prog, [(da, dna, None), (db, dnb, None)], n, [(db, dnb), (da, None), (da, dna)])
    assert m == int(keep.sum())
    assert np.array_equal(_np(outs[0][0]).view(np.int64), b[keep].view(np.int64)) and np.array_equal(_np(outs[1][0]), a[keep])
    assert outs[1][1] is None
    if m:
        assert not D.unpack_nullbits(_np(outs[0][1]).view(np.uint64), m).any()	# the predicate dropped every NULL of these columns
    # a predicate on one column, projection of another with NULLs that survive
    prog1 = [(D.P_CMP_COL_CONST, D.CMP_LE, D.T_INT64, 0, 0, 7)]
    keep1 = (a <= 7) & ~na
    m1, o1 = dev.filter_project(prog1, [(da, dna, None)], n, [(db, dnb)])
    assert m1 == int(keep1.sum())
    if m1:
        assert np.array_equal(_np(o1[0][0]).view(np.int64), b[keep1].view(np.int64))
        assert np.array_equal(D.unpack_nullbits(_np(o1[0][1]).view(np.uint64), m1), nb_[keep1])
    # nothing survives / no projection
    m0, o0 = dev.filter_project([(D.P_CONST, 0, 0, 0, 0, 0)], [], n, [(da, dna)])
    assert m0 == 0 and o0[0][0].numel() == 0


def test_double_join_keys_follow_ieee_equality(dev):
    """DOUBLE equi-join keys (reference cmp_double_value_to_value, executor_select.c:440-460: IEEE `==`): after the
    canonicalisation pass the join's word comparison gives exactly the pairs numpy's float `==` gives - -0.0 joins
    +0.0, no NaN (of any payload or sign) joins anything, NULLs stay NULL."""
    rng = np.random.default_rng(3)
    pool = np.array([0.0, -0.0, np.nan, -np.nan, 0.5, -0.5, 1e308, -1e-308, np.inf, -np.inf])
    specials = np.array([0x7FF8000000000001, 0xFFF0000000000001, 0x7FF0000000000000], dtype=np.uint64).view(np.float64)  # NaNs + inf
    pool = np.concatenate([pool, specials])
    x = pool[rng.integers(0, len(pool), 3001)]
    y = pool[rng.integers(0, len(pool), 2777)]
    xn, yn = rng.random(len(x)) < 0.1, rng.random(len(y)) < 0.1
    kx, nx = dev.double_join_keys(dev.to_dev(x), dev.nullbits_dev(xn))
    ky, ny = dev.double_join_keys(dev.to_dev(y), dev.nullbits_dev(yn))
    pl, pr = dev.join_pairs(kx, nx, ky, ny)
    with np.errstate(invalid="ignore"):
        eq = (x[:, None] == y[None, :]) & ~xn[:, None] & ~yn[None, :]
    el, er = np.nonzero(eq)			# row-major = (left, right) ascending: the reference's emission order
    assert np.array_equal(_np(pl), el) and np.array_equal(_np(pr), er)
    # through a row-id vector (joined streams), and the NULL bits of the result
    idx = rng.integers(0, len(x), 999).astype(np.int32)
    kg, ng = dev.double_join_keys(dev.to_dev(x), dev.nullbits_dev(xn), dev.to_dev(idx))
    gx = x[idx]
    assert np.array_equal(D.unpack_nullbits(_np(ng).view(np.uint64), len(idx)), xn[idx] | np.isnan(gx))
    keep = ~(xn[idx] | np.isnan(gx))
    assert np.array_equal(_np(kg).view(np.float64)[keep], gx[keep] + 0.0)		# x + 0.0 turns -0.0 into +0.0, nothing else
    assert not np.signbit(_np(kg).view(np.float64)[keep & (gx == 0)]).any()


@pytest.mark.parametrize("n_dest", [1, 2, 4, 8])
def test_partition_by_dest(dev, n_dest):
    rng = np.random.default_rng(n_dest)
    k = rng.integers(-10**9, 10**9, 200_001, dtype=np.int64)
    nl = rng.random(len(k)) < 0.03
    ek, ec = orc.partition_by_dest(k, nl, n_dest)
    out, counts = dev.partition_by_dest(dev.to_dev(k), dev.nullbits_dev(nl), n_dest)
    assert counts == ec.tolist()
    got = _np(out)
    # order inside a destination is unspecified: compare each destination's keys as multisets
    off = np.concatenate([[0], np.cumsum(ec)])
    for d in range(n_dest):
        assert np.array_equal(np.sort(got[off[d]:off[d + 1]]), np.sort(ek[off[d]:off[d + 1]]))
    # with row ids: every entry names its source row (that is how payload columns follow the keys)
    out2, counts2, rid = dev.partition_by_dest(dev.to_dev(k), dev.nullbits_dev(nl), n_dest, with_rid=True)
    assert counts2 == counts
    r = _np(rid).view(np.uint32)
    assert np.array_equal(k[r], _np(out2)) and len(np.unique(r)) == len(r) and not nl[r].any()
    assert np.array_equal(orc.dest_of(_np(out2), n_dest), np.repeat(np.arange(n_dest), counts2))


@pytest.mark.parametrize("n_dest", [1, 2, 8])
def test_partition_by_dest_pruned_by_the_other_tables_key_range(dev, n_dest):
    """Min-max pruning before the shuffle: rows whose key lies outside [keep_lo, keep_hi] stay home (with and without the 4-byte
    wire format, with row ids), a key outside the range promised for its own column is an error."""
    rng = np.random.default_rng(50 + n_dest)
    k = rng.integers(-10**6, 10**6, 300_001, dtype=np.int64)
    nl = rng.random(len(k)) < 0.03
    lo, hi = -250_000, 123_456
    inside = (k >= lo) & (k <= hi)
    ek, ec = orc.partition_by_dest(k[inside], nl[inside], n_dest)
    off = np.concatenate([[0], np.cumsum(ec)])
    for keys32 in (False, True):
        out, counts, rid = dev.partition_by_dest(dev.to_dev(k), dev.nullbits_dev(nl), n_dest, with_rid=True, keys32=keys32, keep=(lo, hi),
                                                 own=(-10**6, 10**6))
        assert counts == ec.tolist()
        got = _np(out).astype(np.int64)
        for d in range(n_dest):
            assert np.array_equal(np.sort(got[off[d]:off[d + 1]]), np.sort(ek[off[d]:off[d + 1]]))
        r = _np(rid).view(np.uint32)
        assert np.array_equal(k[r], got) and inside[r].all() and not nl[r].any()
    with pytest.raises(D.DeviceError) as ei:
        dev.partition_by_dest(dev.to_dev(k), dev.nullbits_dev(nl), n_dest, keep=(lo, hi), own=(-10**6, 500_000))
    assert "promised" in str(ei.value)
    out, counts = dev.partition_by_dest(dev.to_dev(k), dev.nullbits_dev(nl), n_dest, keep=(5, 4))	# an empty range keeps nothing
    assert sum(counts) == 0


def test_gen_keys_matches_oracle(dev):
    for (n, first, domain, seed, mod) in [(1000, 0, 1000, 42, 0), (5000, 2500, 10_000, 43, 0), (4096, 0, 4096, 44, 256)]:
        got = _np(dev.gen_keys(n, first, domain, seed, mod))
        assert np.array_equal(got, orc.gen_keys(n, first, domain, seed, mod))
    full = _np(dev.gen_keys(10_000, 0, 10_000, 42, 0))
    assert np.array_equal(np.sort(full), np.arange(10_000))


def test_large_properties_north_star(dev):
    """Size-independent properties at 10^7 rows per table (variant D of SURVEY 8d C3):
    A = permutation of [0,N), B keys = permutation mod N/16  =>  G = N/16 groups of count 16,
    J = N joined rows, groups ordered by first position, keys distinct."""
    N = 10_000_000
    a = dev.gen_keys(N, 0, N, 42, 0)
    b = dev.gen_keys(N, 0, N, 43, N // 16)
    k, c, f, j = dev.join_group_count(a, None, b, None)
    assert j == N
    assert k.numel() == N // 16
    assert bool((c == 16).all())
    assert bool((f[1:] > f[:-1]).all())
    assert bool((a[f.long()] == k).all())
    assert torch.unique(k).numel() == k.numel()
    assert int(k.max()) < N // 16


def test_join_group_count_split_begin_finish(dev):
    """begin() on the left table + finish() with the right table == the one-call operator (the multi-GPU
    pipeline prepares the left table while the right one is still in flight)."""
    rng = np.random.default_rng(77)
    for n_l, n_r, dom in [(1000, 3000, 500), (1_200_000, 900_000, 800_000)]:
        kl = rng.integers(0, dom, n_l, dtype=np.int64)
        kr = rng.integers(0, dom, n_r, dtype=np.int64)
        nl = rng.random(n_l) < 0.02
        ek, ec, ef, ej = orc.join_group_count(kl, nl, kr, None)
        dl, dnl, dr = dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr)
        dev.join_group_count_begin(dl, dnl, n_r + 10)
        k, c, f, j = dev.join_group_count_finish(dr, None)
        assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec) and np.array_equal(_np(f).astype(np.int64), ef)
    # empty right side after a begin
    dev.join_group_count_begin(dev.to_dev(np.arange(10, dtype=np.int64)), None, 5)
    k, c, f, j = dev.join_group_count_finish(dev.to_dev(np.zeros(0, dtype=np.int64)), None)
    assert k.numel() == 0 and j == 0


@pytest.mark.parametrize("n_l,n_r", [(393_216, 393_216), (393_217, 1000), (400_000, 786_433), (786_433, 5), (100, 2_000_000),
                                       (2_000_000, 100), (3_000_000, 3_000_000)])
def test_join_group_count_level_boundaries_and_asymmetric_sizes(dev, n_l, n_r):
    """Sizes around the one-level / two-level switch (393 216 build rows) and very unequal tables
    (the leaf count follows the left table; the right side is streamed whatever its size)."""
    rng = np.random.default_rng(n_l ^ n_r)
    dom = max(10, min(n_l, n_r) * 2)
    kl = rng.integers(0, dom, n_l, dtype=np.int64)
    kr = rng.integers(0, dom, n_r, dtype=np.int64)
    ek, ec, ef, ej = orc.join_group_count(kl, None, kr, None)
    k, c, f, j = dev.join_group_count(dev.to_dev(kl), None, dev.to_dev(kr), None)
    assert j == ej
    assert np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec) and np.array_equal(_np(f).astype(np.int64), ef)


# ---- ORDER BY: stable multi-key sort permutation (extension) -------------------------------------

def _sort_case(dev, rng, n, specs, with_rid=False, topk=None):
    """specs: list of (kind, desc, null_frac); kind in small/full/neg/double/const.  topk: list of k - the top-k operator
    must deliver the first k entries of the same permutation; -> rows it sorted for every k."""
    keys_np, keys_dev, keep = [], [], []
    m = n if not with_rid else 2 * n + 3
    rid = rng.integers(0, m, n).astype(np.uint32) if with_rid else None
    rid_dev = dev.to_dev(rid) if with_rid else None
    for kind, desc, nf in specs:
        if kind == "small":
            v = rng.integers(0, 7, m, dtype=np.int64)
        elif kind == "full":
            v = rng.integers(np.iinfo(np.int64).min, np.iinfo(np.int64).max, m, dtype=np.int64)
        elif kind == "neg":
            v = rng.integers(-1000, 1000, m, dtype=np.int64)
        elif kind == "const":
            v = np.full(m, 42, dtype=np.int64)
        else:
            v = np.round(rng.normal(0, 100, m), 1)
            v[rng.random(m) < 0.05] = -0.0
            v[rng.random(m) < 0.05] = 0.0
            if kind == "nan":          # NaNs of both signs (they can enter through the bulk append API): ordered by their bits, like everything
                v[rng.random(m) < 0.001] = np.nan
                neg = rng.random(m) < 0.0005
                v.view(np.uint64)[neg] = np.uint64(0xFFF8000000000001)
        nulls = (rng.random(m) < nf) if nf else None
        vd = dev.to_dev(v)
        nd = dev.nullbits_dev(nulls)
        keep += [vd, nd]
        keys_np.append((v, nulls, rid, kind in ("double", "nan"), desc))
        keys_dev.append((vd, nd, rid_dev, D.T_DOUBLE if kind in ("double", "nan") else D.T_INT64, desc))
    want = orc.sort_perm(keys_np, n)
    if topk is not None:
        sorted_rows = []
        for k in topk:
            got, cand = dev.topk_perm(keys_dev, n, k)
            assert np.array_equal(_np(got).view(np.uint32), want[:k]), (k, cand)
            sorted_rows.append(cand)
        return sorted_rows
    got = _np(dev.sort_perm(keys_dev, n)).view(np.uint32)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("specs", [
    [("full", False, 0.0)], [("full", True, 0.2)], [("full", False, 0.2)], [("double", False, 0.0)], [("double", True, 0.1)],
    [("neg", False, 0.0), ("full", True, 0.0)], [("neg", True, 0.01), ("double", False, 0.3), ("small", False, 0.0)],
    [("full", False, 0.999)], [("full", True, 0.999)], [("nan", False, 0.0)], [("nan", True, 0.05), ("small", False, 0.0)],
], ids=lambda s: "+".join(f"{k}{'D' if d else 'A'}{int(nf * 10)}" for k, d, nf in s))
def test_topk_perm_is_the_prefix_of_the_stable_sort_unpinned(dev, specs):
    """ORDER BY ... LIMIT k (extension): threshold from a sample, one filter pass, sort of the candidates -
    the same first k positions as the full stable sort for INT64 / DOUBLE first keys, ascending and descending, NULLs
    first / last (also when almost every row is NULL and the threshold is one), ties on the first key broken by
    further keys, through row-id vectors; and only a fraction of the rows goes through the sort."""
    rng = np.random.default_rng(len(specs) * 11 + 3)
    n = 400_000
    ks = [1, 10, 1000, n // 40]
    for with_rid in (False, True):
        cand = _sort_case(dev, rng, n, specs, with_rid=with_rid, topk=ks)
        if specs[0][0] != "neg" and specs[0][2] < 0.9:
            assert all(c < n // 4 for c in cand), cand


@pytest.mark.parametrize("specs,n", [([("small", False, 0.0)], 300_000), ([("const", True, 0.0), ("full", False, 0.0)], 300_000),
                                     ([("full", False, 0.0)], 5000), ([("full", True, 0.0)], 0)])
def test_topk_perm_falls_back_to_the_full_sort_unpinned(dev, specs, n):
    """few distinct first-key values (every candidate set is most of the table), small inputs, k above n/8, k > n"""
    rng = np.random.default_rng(n + 1)
    cand = _sort_case(dev, rng, n, specs, topk=[3, 50, n // 2, n, n + 5] if n else [0, 3])
    assert n == 0 or cand[-1] == n


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 4095, 4096, 4097, 100_000, 1_000_003])
def test_sort_perm_single_key_sizes_unpinned(dev, n):
    rng = np.random.default_rng(n)
    _sort_case(dev, rng, n, [("neg", False, 0.0)])
    _sort_case(dev, rng, n, [("small", True, 0.1)])


@pytest.mark.parametrize("specs", [
    [("full", False, 0.0)], [("full", True, 0.2)], [("double", False, 0.0)], [("double", True, 0.1)], [("const", False, 0.0)],
    [("const", True, 0.5)], [("small", False, 0.0), ("neg", True, 0.0)], [("small", True, 0.3), ("double", False, 0.3), ("small", False, 0.0)],
    [("small", False, 1.0)], [("neg", False, 0.0), ("full", False, 0.0), ("double", True, 0.0), ("small", True, 0.2)],
], ids=lambda s: "+".join(f"{k}{'D' if d else 'A'}{int(nf * 10)}" for k, d, nf in s))
def test_sort_perm_key_kinds_unpinned(dev, specs):
    rng = np.random.default_rng(len(specs) * 7 + 1)
    _sort_case(dev, rng, 50_000, specs)
    _sort_case(dev, rng, 50_000, specs, with_rid=True)


def test_sort_perm_large_property_unpinned(dev):
    """10^7 rows, 2 keys: result is a permutation, keys come out non-decreasing, ties keep stream order."""
    n = 10_000_000
    a = dev.gen_keys(n, 0, n, 5, 1000)			# 1000 distinct values
    b = dev.gen_keys(n, 0, n, 6, 0)			# permutation
    perm = dev.sort_perm([(a, None, None, D.T_INT64, False), (b, None, None, D.T_INT64, True)], n).to(torch.int64)
    assert torch.equal(torch.sort(perm).values, torch.arange(n, device=perm.device))
    sa, sb = a[perm], b[perm]
    assert bool((sa[1:] >= sa[:-1]).all())
    same = sa[1:] == sa[:-1]
    assert bool((sb[1:][same] <= sb[:-1][same]).all())
    one = dev.sort_perm([(a, None, None, D.T_INT64, False)], n).to(torch.int64)
    s1 = a[one]
    tie = s1[1:] == s1[:-1]
    assert bool((one[1:][tie] > one[:-1][tie]).all())	# stability


@pytest.mark.parametrize("n", [1, 2, 64, 4097, 60_000])
def test_distinct_sel_unpinned(dev, n):
    rng = np.random.default_rng(n + 11)
    a = rng.integers(-3, 4, n, dtype=np.int64)
    b = np.round(rng.normal(0, 1, n), 0)
    na = rng.random(n) < 0.2
    rid = rng.integers(0, n, n).astype(np.uint32)
    ad, bd, nad, ridd = dev.to_dev(a), dev.to_dev(b), dev.nullbits_dev(na), dev.to_dev(rid)
    for keys_np, keys_dev in [
            ([(a, None, None, False, False)], [(ad, None, None, D.T_INT64, False)]),
            ([(a, na, None, False, False), (b, None, None, True, False)], [(ad, nad, None, D.T_INT64, False), (bd, None, None, D.T_DOUBLE, False)]),
            ([(b, None, rid, True, False), (a, na, rid, False, False)], [(bd, None, ridd, D.T_DOUBLE, False), (ad, nad, ridd, D.T_INT64, False)])]:
        got = _np(dev.distinct_sel(keys_dev, n)).view(np.uint32)
        assert np.array_equal(got, orc.distinct_sel(keys_np, n))


@pytest.mark.parametrize("variant", ["D", "U"])
def test_full_size_north_star_properties(dev, variant):
    """BASELINE config 3 at its full size (10^8 rows per table) through size-independent properties:
    D: B keys = permutation mod N/16 -> G = N/16 groups of exactly 16, J = N; U: both permutations -> G = N groups
    of 1.  Keys distinct, groups in first-occurrence order of the left table, key = a[first], and the closed-form
    checksums sum(keys) / sum(counts)."""
    N = 100_000_000
    a = dev.gen_keys(N, 0, N, 42, 0)
    b = dev.gen_keys(N, 0, N, 43, N // 16 if variant == "D" else 0)
    k, c, f, j = dev.join_group_count(a, None, b, None)
    G = N // 16 if variant == "D" else N
    assert j == N and k.numel() == G
    assert bool((c == (16 if variant == "D" else 1)).all()) and int(c.sum()) == N
    assert bool((f[1:] > f[:-1]).all())
    assert bool((a[f.long()] == k).all())
    assert int(k.sum()) == G * (G - 1) // 2 and int(k.min()) == 0 and int(k.max()) == G - 1	# a permutation of [0, G)
    del k, c, f
    torch.cuda.empty_cache()


@pytest.mark.parametrize("variant", ["D", "U", "S"])
def test_full_size_north_star_in_any_order(dev, variant):
    """The same full-size workloads through the operator WITHOUT MDB_ORDER_FIRST (no row ids, no ordering sort; U and S through the
    4096-digit first level): the groups as a set - distinct keys, the closed-form key and count sums, every count what the generator
    makes it - and, key for key, the ordered operator's result (both sorted by key on the device)."""
    N = 100_000_000
    a = dev.gen_keys(N, 0, N, 42, 0)
    b = dev.gen_keys(N, 0, N, 43, 0 if variant == "U" else N // 16)
    if variant == "S":
        b.mul_(16)
    k, c, j = dev.join_group_count_unordered(a, None, b, None)
    assert dev.last_join_unordered()
    G = N if variant == "U" else N // 16
    assert j == N and k.numel() == G and int(c.sum()) == N
    assert bool((c == (1 if variant == "U" else 16)).all())
    ks, order = torch.sort(k)
    assert bool((ks[1:] > ks[:-1]).all())                                              # every key once
    step = 16 if variant == "S" else 1
    assert int(ks[0]) == 0 and int(ks[-1]) == step * (G - 1) and int(ks.sum()) == step * G * (G - 1) // 2
    cs = c[order]
    del k, c, order
    ko, co, fo, jo = dev.join_group_count(a, None, b, None)
    assert jo == j and ko.numel() == G
    kos, oo = torch.sort(ko)
    assert bool((kos == ks).all()) and bool((co[oo] == cs).all())
    del ko, co, fo, kos, oo, ks, cs
    torch.cuda.empty_cache()


def test_full_size_three_way_join_properties(dev):
    """BASELINE config 5 shape at 10^8 rows per table on one GPU (keys only): A, B, C independent permutations of
    [0, N): (A join B) join C has exactly N rows, every A row once, and the composed row ids point at equal keys."""
    N = 100_000_000
    a, b, cc = (dev.gen_keys(N, 0, N, s, 0) for s in (42, 43, 44))
    l, r = dev.join_pairs(a, None, b, None)
    assert l.numel() == N
    assert bool((l.long() == torch.arange(N, device=l.device)).all())		# left-major order, 1:1
    assert bool((b[r.long()] == a).all())
    del l
    p, q = dev.join_pairs(a, None, cc, None)
    assert p.numel() == N and bool((cc[q.long()] == a).all())
    first, cnt = dev.group_count(a, None)
    assert first.numel() == N and bool((cnt == 1).all())
    del first, cnt
    # DOUBLE / INT payload carried through both joins (x of A, y of B, z of C: SURVEY 8d C5), bit-exact: the projection of the
    # three-way join gathers y through r and z through q; every gathered cell must be the cell of the row with the equal key
    y = dev.gen_payload(N, 0, 143, 1)
    z = dev.gen_payload(N, 0, 144, 0)
    (yo, _), (zo, _) = dev.gather_cols([(y, None, r), (z, None, q)], N)
    # inverse check without a host copy: scatter the gathered cells back by the partner row - the source column must re-appear
    back = torch.empty_like(y)
    back[r.long()] = yo
    assert bool((back.view(torch.int64) == y.view(torch.int64)).all())
    backz = torch.empty_like(z)
    backz[q.long()] = zo
    assert bool((backz == z).all())
    del p, q, r, y, z, yo, zo, back, backz
    # ... and the same query as ONE operator: every key once, COUNT(*) = 1, first rows = A's rows, 10^8 joined rows
    k3, c3, f3, j3 = dev.join_group_count_multi(a, None, [(b, None), (cc, None)])
    assert j3 == N and k3.numel() == N and bool((c3 == 1).all()) and bool((k3 == a).all())
    assert bool((f3.long() == torch.arange(N, device=f3.device)).all())
    assert dev.last_join_multi()
    del k3, c3, f3
    # ... and in any order: the same set of groups
    ku, cu, ju = dev.join_group_count_multi_unordered(a, None, [(b, None), (cc, None)])
    assert dev.last_join_unordered() and ju == N and ku.numel() == N and bool((cu == 1).all())
    assert bool((torch.sort(ku)[0] == torch.arange(N, device=ku.device)).all())
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n", [2_097_153, 3_000_000, 20_000_000, 67_108_865])
def test_group_count_all_unique_partial_id_range(dev, n):
    """GROUP BY over all-unique keys (G = n): every row id is a group's first row, and n just above a power of two
    leaves half of the ordering sort's first-level digits empty - the fixed-capacity regions must be sized for the
    digits that occur (regression: overflowing regions silently dropped records)."""
    a = dev.gen_keys(n, 0, n, 42, 0)
    first, cnt = dev.group_count(a, None)
    assert first.numel() == n
    assert bool((cnt == 1).all())
    assert bool((first.long() == torch.arange(n, device=first.device)).all())


@pytest.mark.parametrize("seed", range(10))
def test_join_group_count_randomised_large(dev, seed):
    """Randomised sizes / key domains / duplicate factors / NULL rates up to 3*10^7 rows against the C hash-join
    oracle (oracle/cpu_hash.c, pinned to the reference vectors): exact keys, counts, first rows, joined rows, order."""
    from oracle import cpu
    rng = np.random.default_rng(1000 + seed)
    n_l = int(10 ** rng.uniform(5, 7.48))
    n_r = int(10 ** rng.uniform(5, 7.48))
    dom = int(max(2, n_l * 10 ** rng.uniform(-2.5, 0.5)))
    lo = int(rng.integers(-dom, dom))
    kl = rng.integers(lo, lo + dom, n_l, dtype=np.int64)
    kr = rng.integers(lo, lo + dom, n_r, dtype=np.int64)
    if seed % 3 == 0:
        kl = rng.permutation(n_l).astype(np.int64) + lo		# unique build keys: G close to n_l
    nl = (rng.random(n_l) < 0.02) if seed % 2 else None
    nr = (rng.random(n_r) < 0.02) if seed % 4 == 1 else None
    ek, ec, ef, ej = cpu.hash_join_group_count(kl, nl, kr, nr, 8)
    k, c, f, j = dev.join_group_count(dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr), dev.nullbits_dev(nr))
    assert j == ej and k.numel() == len(ek)
    assert np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec) and np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef)
    # plain GROUP BY over the left keys: counts per key in first-occurrence order (numpy oracle)
    first, cnt = dev.group_count(dev.to_dev(kl), dev.nullbits_dev(nl))
    e_first, e_cnt = orc.group_count(kl, nl)
    assert np.array_equal(_np(first).view(np.uint32).astype(np.int64), e_first) and np.array_equal(_np(cnt), e_cnt)


def test_join_pairs_two_level_with_a_multi_chunk_leaf(dev):
    """Two partition levels (histogram-free layout first) plus one key with 5000 right rows: that leaf spans several
    emit chunks, so the operator has to notice and redo the join with the right side in row-id order."""
    rng = np.random.default_rng(12)
    n_l, n_r = 600_000, 1_000_000
    kl = rng.integers(0, 800_000, n_l, dtype=np.int64)
    kr = rng.integers(0, 800_000, n_r, dtype=np.int64)
    kl[rng.choice(n_l, 300, replace=False)] = 77
    kr[rng.choice(n_r, 5000, replace=False)] = 77
    el, er = orc.join_pairs(kl, None, kr, None)
    l, r = dev.join_pairs(dev.to_dev(kl), None, dev.to_dev(kr), None)
    assert l.numel() == len(el)
    assert np.array_equal(_np(l).astype(np.int64), el) and np.array_equal(_np(r).astype(np.int64), er)


@pytest.mark.parametrize("seed", range(6))
def test_join_pairs_randomised_large(dev, seed):
    rng = np.random.default_rng(300 + seed)
    n_l = int(10 ** rng.uniform(5, 6.6))
    n_r = int(10 ** rng.uniform(5, 6.6))
    dom = int(max(2, max(n_l, n_r) * 10 ** rng.uniform(-0.5, 0.7)))	# about 0.2 .. 3 matches per row
    kl, nl = _mk(rng, n_l, dom, 0.02 if seed % 2 else 0.0, -5)
    kr, nr = _mk(rng, n_r, dom, 0.02 if seed % 3 == 0 else 0.0, -5)
    el, er = orc.join_pairs(kl, nl, kr, nr)
    l, r = dev.join_pairs(dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr), dev.nullbits_dev(nr))
    assert l.numel() == len(el)
    assert np.array_equal(_np(l).astype(np.int64), el) and np.array_equal(_np(r).astype(np.int64), er)


@pytest.mark.parametrize("case", ["tiny", "unique_2e22", "dups_both", "nulls", "sparse_2e29", "beyond_any_window", "empty_right", "no_match"])
def test_join_keys_is_the_key_column_of_join_pairs_as_a_multiset(dev, case):
    """mdb_dev_join_keys (BASELINE configs[3]: a join whose only output is its key columns): every joined row's key, in unspecified
    order - equal to keys_l[pos_l] over the oracle's pairs as a multiset; every size class (single-workgroup kernel, one level, the
    4096-digit level, two levels, the ordered operator as the fallback)"""
    rng = np.random.default_rng(hash(case) % 1000)
    nl = nr = None
    if case == "tiny":
        kl, kr = rng.integers(0, 20, 300), rng.integers(0, 20, 200)
    elif case == "unique_2e22":
        kl, kr = rng.permutation(3_000_000) + 50, rng.permutation(3_000_000)[:2_500_000] + 50
    elif case == "dups_both":
        kl, kr = rng.integers(-1000, 400_000, 1_500_000), rng.integers(-1000, 400_000, 1_200_000)
    elif case == "nulls":
        kl, kr = rng.integers(0, 2_000_000, 1_300_000), rng.integers(0, 2_000_000, 1_300_000)
        nl, nr = rng.random(len(kl)) < 0.05, rng.random(len(kr)) < 0.05
    elif case == "sparse_2e29":
        pool = rng.integers(0, 1 << 29, 1_000_000)
        kl, kr = pool[rng.integers(0, len(pool), 1_400_000)] + 10**12, pool[rng.integers(0, len(pool), 1_100_000)] + 10**12
    elif case == "beyond_any_window":
        pool = rng.integers(-2**62, 2**62, 800_000)
        kl, kr = pool[rng.integers(0, len(pool), 1_200_000)], pool[rng.integers(0, len(pool), 1_200_000)]
    elif case == "empty_right":
        kl, kr = rng.integers(0, 1000, 5000), np.zeros(0, dtype=np.int64)
    else:
        kl, kr = rng.integers(0, 10**6, 1_200_000), rng.integers(2 * 10**6, 3 * 10**6, 1_200_000)
    kl, kr = kl.astype(np.int64), kr.astype(np.int64)
    got = dev.join_keys(dev.to_dev(kl), dev.nullbits_dev(nl) if nl is not None else None, dev.to_dev(kr), dev.nullbits_dev(nr) if nr is not None else None)
    el, er = orc.join_pairs(kl, nl, kr, nr)
    assert got.numel() == len(el)
    assert np.array_equal(np.sort(_np(got)), np.sort(kl[el]))


@pytest.mark.parametrize("case", ["pk_pk_two_cells", "fk_to_pk_one_cell", "window_far_from_zero", "left_row_without_partner", "null_left_key",
                                  "duplicate_right_key", "keys_beyond_a_window", "small", "window_2e27_two_levels", "window_2e29_two_cells",
                                  "window_2e27_duplicate_right_key", "row_order_2e25", "row_order_2e26_two_cells", "row_order_whole_tiles",
                                  "row_order_left_row_without_partner", "row_order_hot_key", "row_order_window_2e16", "row_order_window_2e19_two_cells"])
def test_join_payload_carries_the_right_tables_cells_to_every_left_row(dev, case, monkeypatch):
    """mdb_dev_join_payload (BASELINE configs[1]: a primary-key join with payload): when every left row has exactly one partner the
    outputs are the partners' payload cells in left-row order (INT64 and DOUBLE bits alike) - equal to payload[pos_r] over the oracle's
    pairs; any other join is refused with "not served" (the pairs path answers), never answered wrongly"""
    rng = np.random.default_rng(len(case) * 13 + 5)
    n = 1_500_000
    nl = None
    served = True
    if case in ("pk_pk_two_cells", "window_far_from_zero", "left_row_without_partner", "null_left_key", "duplicate_right_key"):
        off = 10**12 if case == "window_far_from_zero" else 0
        kl, kr = rng.permutation(n).astype(np.int64) + off, rng.permutation(n).astype(np.int64) + off
        if case == "left_row_without_partner":
            kl[12345] = n + 77 + off
            served = False
        if case == "null_left_key":
            nl = np.zeros(n, dtype=bool)
            nl[777] = True
            served = False
        if case == "duplicate_right_key":
            kr[5] = kr[6]
            served = False
        pay = [rng.integers(-2**60, 2**60, n, dtype=np.int64), rng.standard_normal(n)]
    elif case == "fk_to_pk_one_cell":		# a fact table's foreign key: 3 * 10^6 left rows over 2 * 10^5 right keys
        kr = rng.permutation(200_000).astype(np.int64) + 1000
        kl = kr[rng.integers(0, len(kr), 3_000_000)]
        pay = [rng.integers(0, 10**9, len(kr), dtype=np.int64)]
    elif case in ("window_2e27_two_levels", "window_2e29_two_cells", "window_2e27_duplicate_right_key"):
        # sparse unique keys over 2^27 / 2^29 values: two partition levels, the cells travel through both
        span = 1 << (29 if "2e29" in case else 27)
        kr = np.unique(rng.integers(0, span, 2_200_000, dtype=np.int64)) - 12345
        kl = kr[rng.integers(0, len(kr), 2_500_000)]		# a foreign key: duplicates on the left
        pay = [rng.integers(-2**62, 2**62, len(kr), dtype=np.int64)] + ([rng.standard_normal(len(kr))] if "two_cells" in case else [])
        if "duplicate" in case:
            kr = kr.copy()
            kr[100] = kr[101]
            served = False
    elif case.startswith("row_order"):
        # round 5 (mdb_dev_rowjoin.hip): windows of 2^25 ... 2^27 values - the left table sorted tile by tile, the result placed in row order
        # (the operator pads the sampled range and rounds it up to a power of two: these spans give windows of 2^25, 2^26, 2^27)
        # (2e16 / 2e19: a fact table against a small dimension - few digits, pieces of hundreds of words: the leaf's whole-wave walk and
        # its one-piece-per-instruction form)
        span = 1 << (24 if "2e25" in case else 25 if "2e26" in case else 15 if "2e16" in case else 18 if "2e19" in case else 26)
        kr = np.unique(rng.integers(0, span, min(1_800_000, span), dtype=np.int64)) + 10**10
        nleft = 32768 * 70 if "whole_tiles" in case else 2_345_679
        kl = kr[rng.integers(0, len(kr), nleft)]
        if "hot_key" in case:		# one key on 200 000 consecutive left rows: pieces of a whole tile in one digit
            kl[500_000:700_000] = kr[4242]
        if "without_partner" in case:
            kl[nleft - 3] = kr[7] + 1 if kr[7] + 1 != kr[8] else kr[7] - 1
            if kl[nleft - 3] in (kr[6], kr[8]):
                kl[nleft - 3] = 10**10 + span - 1 if kr[-1] != 10**10 + span - 1 else 10**10
            served = False
        pay = [rng.integers(-2**62, 2**62, len(kr), dtype=np.int64)] + ([rng.standard_normal(len(kr))] if "two_cells" in case else [])
    elif case == "keys_beyond_a_window":
        kr = np.unique(rng.integers(-2**62, 2**62, n, dtype=np.int64))
        kl = rng.permutation(kr)
        pay = [np.arange(len(kr), dtype=np.int64)]
        served = False
    else:
        kl, kr = rng.permutation(3000).astype(np.int64), rng.permutation(3000).astype(np.int64)
        pay = [np.arange(3000, dtype=np.int64)]
        served = False			# (small tables: the pairs path has fewer launches)
    dev.prof_enable(True)
    dev.prof_reset()
    if case.startswith("row_order") or case in ("pk_pk_two_cells", "fk_to_pk_one_cell"):
        monkeypatch.setenv("MDB_ROWJOIN", "2")     # (by default the form takes left tables of 2^24 rows and more)
    got = dev.join_payload(dev.to_dev(kl), dev.nullbits_dev(nl) if nl is not None else None, dev.to_dev(kr), None, [dev.to_dev(p) for p in pay])
    ran = set(dev.prof_read())
    dev.prof_enable(False)
    if case in ("pk_pk_two_cells", "fk_to_pk_one_cell"):   # small windows: fewer digits, longer pieces (other instances of the leaf kernel)
        assert "rowjoin_leaf" in ran, (case, ran)
    if case.startswith("row_order"):
        assert {"rowjoin_tile_sort", "rowjoin_leaf"} <= ran and (not served or "rowjoin_place" in ran), (case, ran)
    if not served:
        assert got is None, case
        return
    assert got is not None, case
    el, er = orc.join_pairs(kl, None, kr, None)
    assert len(el) == len(kl) and np.array_equal(el, np.arange(len(kl)))
    for g, p in zip(got, pay):
        assert np.array_equal(_np(g).view(np.int64), p[er].view(np.int64)), case
    # and the same call again (remembered verdicts); the default choice of form this time
    if "hot_key" not in case:       # (the older forms' fixed-capacity regions overflow under a hot key: "not served", by design)
        monkeypatch.delenv("MDB_ROWJOIN", raising=False)
    again = dev.join_payload(dev.to_dev(kl), None, dev.to_dev(kr), None, [dev.to_dev(p) for p in pay])
    assert again is not None and np.array_equal(_np(again[0]).view(np.int64), pay[0][er].view(np.int64))


@pytest.mark.parametrize("case", ["two_tables_one_cell_each", "two_tables_two_cells_each", "three_tables", "one_table_two_cells", "window_2e19",
                                  "left_row_without_partner_in_the_second_table", "duplicate_key_in_the_second_table", "key_outside_the_bound",
                                  "foreign_keys_on_the_left", "partial_last_tile_only"])
def test_join_payload_multi_one_left_sort_serves_every_right_table(dev, case, monkeypatch):
    """mdb_dev_join_payload_multi (BASELINE configs[4]'s join-only form: SELECT * over A, B, C on one key; the reference joins B, then C
    against the materialised A x B - executor_select.c:1076-1232): right[t].out[c][i] = payload cell c of left row i's partner in table
    t, bit for bit what payload[np_oracle.join_pairs] says; the left table's tiles are sorted ONCE, one leaf launch and one placement
    pass serve all tables; any statement about the tables that does not hold is refused ("not served"), never answered wrongly"""
    rng = np.random.default_rng(len(case) * 7 + 1)
    monkeypatch.setenv("MDB_ROWJOIN", "2")     # (by default the form takes left tables of 2^24 rows and more)
    span = 1 << (18 if case == "window_2e19" else 21)
    base = -5_000_000
    ntab = 3 if case == "three_tables" else 1 if case == "one_table_two_cells" else 2
    cells = 2 if "two_cells" in case else 1
    served = True
    nleft = 20_000 if case == "partial_last_tile_only" else 1_234_567
    rights = []
    for t in range(ntab):
        kr = rng.permutation(span).astype(np.int64)[: span - 1000 * t] + base     # (tables of different sizes: different tile counts)
        if case == "partial_last_tile_only":
            kr = kr[:30_000]
        pay = [rng.integers(-2**62, 2**62, len(kr), dtype=np.int64), rng.standard_normal(len(kr))][:cells]
        rights.append((kr, pay))
    common = rights[0][0]
    for kr, _ in rights[1:]:
        common = np.intersect1d(common, kr)
    kl = common[rng.integers(0, len(common), nleft)] if case in ("foreign_keys_on_the_left", "partial_last_tile_only") else rng.permutation(common)[:nleft]
    lo, hi = base, base + span - 1
    if case == "left_row_without_partner_in_the_second_table":
        missing = np.setdiff1d(rights[0][0], rights[1][0])
        kl = kl.copy()
        kl[nleft // 2] = missing[0]
        served = False
    if case == "duplicate_key_in_the_second_table":
        kr = rights[1][0].copy()
        kr[10] = kr[11]
        rights[1] = (kr, rights[1][1])
        kl = kl[(kl != rights[1][0][10])]
        served = False
    if case == "key_outside_the_bound":
        kl = kl.copy()
        kl[5] = hi + 3
        served = False
    dev.prof_enable(True)
    dev.prof_reset()
    got = dev.join_payload_multi(dev.to_dev(kl), [(dev.to_dev(kr), [dev.to_dev(p) for p in pay]) for kr, pay in rights], lo, hi)
    ran = dev.prof_read()
    dev.prof_enable(False)
    plan = dev.last_plan()
    nfull, npart = len(kl) // 32768, 1 if len(kl) % 32768 else 0
    assert ran["rowjoin_tile_sort"][0] == (1 if nfull else 0) + npart, ran      # the left table: once, whatever the number of right tables
    assert ran["rowjoin_leaf"][0] == 1 and (not served or ran["rowjoin_place"][0] == 1), ran
    if not served:
        assert got is None and plan["payload_form"] == 0, (case, plan)
        return
    assert got is not None and plan["payload_form"] == 3 and plan["payload_tables"] == ntab and plan["samples"] == 0, (case, plan)
    for (kr, pay), outs in zip(rights, got):
        el, er = orc.join_pairs(kl, None, kr, None)
        assert np.array_equal(el, np.arange(len(kl)))
        for g, p in zip(outs, pay):
            assert np.array_equal(_np(g).view(np.int64), p[er].view(np.int64)), case


@pytest.mark.parametrize("seed", range(8))
def test_join_payload_multi_random_shapes(dev, seed, monkeypatch):
    """mdb_dev_join_payload_multi over seeded random shapes - 1 to 4 right tables, 1 or 2 cells each (4 columns at most), windows of 2^15 ...
    2^22 values anywhere in int64, left tables from a fraction of a tile to dozens of tiles, foreign keys or a permutation on the left:
    every carried column equals payload[np_oracle.join_pairs], bit for bit"""
    rng = np.random.default_rng(9000 + seed)
    monkeypatch.setenv("MDB_ROWJOIN", "2")
    bits = int(rng.integers(15, 23))
    span = 1 << bits
    base = int(rng.integers(-2**40, 2**40))
    ntab = int(rng.integers(1, 5))
    cells = [int(rng.integers(1, 3)) for _ in range(ntab)]
    while sum(cells) > 4:
        cells[int(np.argmax(cells))] -= 1
    universe = rng.permutation(span).astype(np.int64)[: int(span * rng.uniform(0.5, 1.0))] + base
    rights = []
    for t in range(ntab):
        kr = rng.permutation(universe)
        pay = [rng.integers(-2**63, 2**63 - 1, len(kr), dtype=np.int64), rng.standard_normal(len(kr))][: cells[t]]
        rights.append((kr, pay))
    nleft = int(rng.integers(1000, 40 * 32768))
    kl = universe[rng.integers(0, len(universe), nleft)] if seed % 2 else rng.permutation(universe)[: min(nleft, len(universe))]
    got = dev.join_payload_multi(dev.to_dev(kl), [(dev.to_dev(kr), [dev.to_dev(p) for p in pay]) for kr, pay in rights], base, base + span - 1)
    plan = dev.last_plan()
    assert got is not None and plan["payload_form"] == 3 and plan["payload_tables"] == ntab, (seed, plan)
    for (kr, pay), outs in zip(rights, got):
        order = np.argsort(kr)
        pos = order[np.searchsorted(kr[order], kl)]
        for g, p in zip(outs, pay):
            assert np.array_equal(_np(g).view(np.int64), p[pos].view(np.int64)), seed


@pytest.mark.parametrize("persist", ["0", "1"])
@pytest.mark.parametrize("shape", ["hot_key_on_the_left", "window_2e27", "right_table_smaller_than_a_tile", "nullable_right_key_column"])
def test_join_payload_multi_leaf_grids_and_long_pieces(dev, shape, persist, monkeypatch):
    """the row-order leaf both ways (workgroups that stay and walk their XCD's digits / one workgroup per digit: MDB_RJ_PERSIST) over what its piece
    walker finds hardest: a key on 150 000 consecutive left rows (pieces of a whole tile: thousands of tiers), a window of 2^27 values (8192 digits,
    pieces of a few words, most sweeps with a long piece), a right table of a fraction of a tile; and a right key column that comes with a NULL
    bitmap is refused, not read"""
    rng = np.random.default_rng(31 + len(shape))
    monkeypatch.setenv("MDB_ROWJOIN", "2")
    monkeypatch.setenv("MDB_RJ_PERSIST", persist)
    if shape == "window_2e27":
        span, nr, nl = 1 << 27, 3_000_000, 2_500_000
    elif shape == "right_table_smaller_than_a_tile":
        span, nr, nl = 1 << 16, 9_000, 400_000
    else:
        span, nr, nl = 1 << 20, 700_000, 1_000_000
    base = 123_456_789
    kr = rng.choice(span, nr, replace=False).astype(np.int64) + base
    kr2 = rng.permutation(kr)
    kl = kr[rng.integers(0, nr, nl)]
    if shape == "hot_key_on_the_left":
        kl[200_000:350_000] = kr[99]
    pay1, pay2 = rng.integers(-2**62, 2**62, nr, dtype=np.int64), rng.standard_normal(nr)
    rights = [(kr, [pay1]), (kr2, [pay2])]
    if shape == "nullable_right_key_column":
        arr = (D.PayloadRight * 1)()
        d_kr, d_p = dev.to_dev(kr), dev.to_dev(pay1)
        nb = dev.nullbits_dev(np.zeros(nr, dtype=bool))
        out = torch.empty(nl, dtype=torch.int64, device=dev.device)
        arr[0].keys, arr[0].nulls, arr[0].rows, arr[0].npay = d_kr.data_ptr(), nb.data_ptr(), nr, 1
        arr[0].pay_in[0], arr[0].out[0] = d_p.data_ptr(), out.data_ptr()
        d_kl = dev.to_dev(kl)
        rc = dev.lib.mdb_dev_join_payload_multi(dev.h, d_kl.data_ptr(), None, nl, arr, 1, base, base + span - 1)
        assert rc == 1
        return
    got = dev.join_payload_multi(dev.to_dev(kl), [(dev.to_dev(k), [dev.to_dev(p) for p in pay]) for k, pay in rights], base, base + span - 1)
    assert got is not None and dev.last_plan()["payload_tables"] == 2
    for (k, pay), outs in zip(rights, got):
        order = np.argsort(k)
        pos = order[np.searchsorted(k[order], kl)]
        assert np.array_equal(_np(outs[0]).view(np.int64), pay[0][pos].view(np.int64)), (shape, persist)


def test_join_group_count_huge_count_takes_the_dense_ordering(dev):
    """A COUNT(*) that does not fit beside its row id in a 64-bit group record (2^28 > n_l > 2^27 -> 28 id bits, so
    counts >= 2^36): flagged by the leaf kernel, the operator redoes the query with the dense ordering."""
    n_l, dup = (1 << 27) + 5, 300_000
    kl = torch.arange(n_l, dtype=torch.int64, device=dev.device) + 1_000_000
    kl[:dup] = 7
    kr = torch.cat([torch.full((dup,), 7, dtype=torch.int64, device=dev.device),
                    torch.arange(1_000_000 + dup, 1_000_000 + dup + 1000, dtype=torch.int64, device=dev.device)])
    k, c, f, j = dev.join_group_count(kl, None, kr, None)
    assert k.numel() == 1001 and j == dup * dup + 1000
    assert int(k[0]) == 7 and int(c[0]) == dup * dup and int(f[0]) == 0
    assert bool((c[1:] == 1).all()) and bool((k[1:] == torch.arange(1_000_000 + dup, 1_000_000 + dup + 1000, device=dev.device)).all())
    assert bool((f[1:].long() == torch.arange(dup, dup + 1000, device=dev.device)).all())
    del kl, kr
    torch.cuda.empty_cache()


@pytest.mark.parametrize("packed", ["1", "0"])
@pytest.mark.parametrize("n", [1, 2, 65, 4097, 80_000])
def test_group_count_multi_unpinned(dev, n, packed, monkeypatch):
    """packed = 1: the columns' composite value as ONE key column through the single-column operator (round 6); 0: the sort of the stream."""
    monkeypatch.setenv("MDB_GROUP_MULTI_PACKED", packed)
    rng = np.random.default_rng(n + 21)
    a = rng.integers(-2, 3, n, dtype=np.int64)
    b = rng.integers(0, 4, n, dtype=np.int64)
    x = np.round(rng.normal(0, 1, n), 0)
    na, nx = rng.random(n) < 0.2, rng.random(n) < 0.1
    rid = rng.integers(0, n, n).astype(np.uint32)
    ad, bd, xd, nad, nxd, ridd = dev.to_dev(a), dev.to_dev(b), dev.to_dev(x), dev.nullbits_dev(na), dev.nullbits_dev(nx), dev.to_dev(rid)
    for keys_np, keys_dev in [
            ([(a, na, None, False, False), (b, None, None, False, False)], [(ad, nad, None, D.T_INT64, False), (bd, None, None, D.T_INT64, False)]),
            ([(x, nx, None, True, False), (a, na, None, False, False), (b, None, None, False, False)],
             [(xd, nxd, None, D.T_DOUBLE, False), (ad, nad, None, D.T_INT64, False), (bd, None, None, D.T_INT64, False)]),
            ([(b, None, rid, False, False), (x, nx, rid, True, False)], [(bd, None, ridd, D.T_INT64, False), (xd, nxd, ridd, D.T_DOUBLE, False)])]:
        first, cnt = dev.group_count_multi(keys_dev, n)
        ef, ec = orc.group_count_multi(keys_np, n)
        assert np.array_equal(_np(first).view(np.uint32).astype(np.int64), ef) and np.array_equal(_np(cnt), ec)


@pytest.mark.parametrize("shape", ["wide_int64", "two_doubles", "63_bits", "64_bits", "unique_pairs", "one_hot_combination", "all_null_column", "five_columns",
                                   "doubles_one_binade"])
def test_group_count_multi_composite_key_forms(dev, shape):
    """GROUP BY over several columns, the composite value handed to the single-column operator where the ranges fit 63 bits together and
    sorted where they do not: ranges at the limit, DOUBLE columns (bits compared), every row its own group, one combination holding half
    the rows, a column of NULLs only, more columns than the packed form takes - against the numpy oracle."""
    n = 300_000
    rng = np.random.default_rng(len(shape))
    I, Dbl = D.T_INT64, D.T_DOUBLE
    if shape == "wide_int64":       # full-range values: the general path
        cols = [(rng.integers(-2**62, 2**62, n, dtype=np.int64) // 2**40 * 2**40, None, False), (rng.integers(-2**62, 2**62, n, dtype=np.int64) // 2**50 * 2**50, None, False)]
    elif shape == "two_doubles":
        cols = [(np.round(rng.normal(0, 2, n), 0), rng.random(n) < 0.05, True), (rng.choice(np.array([-0.0, 0.0, 1.5, -1.5, np.inf, -np.inf]), n), None, True)]
    elif shape == "63_bits":        # 31 + 32 bits
        cols = [(rng.integers(0, 2**31, n, dtype=np.int64) // 2**14 * 2**14, None, False), (rng.integers(0, 2**32, n, dtype=np.int64) // 2**16 * 2**16, None, False)]
        cols[0][0][:2] = [0, 2**31 - 1]
        cols[1][0][:2] = [0, 2**32 - 1]
    elif shape == "64_bits":        # 32 + 32 bits: one too many
        cols = [(rng.integers(0, 2**32, n, dtype=np.int64) // 2**15 * 2**15, None, False), (rng.integers(0, 2**32, n, dtype=np.int64) // 2**16 * 2**16, None, False)]
        cols[0][0][:2] = [0, 2**32 - 1]
        cols[1][0][:2] = [0, 2**32 - 1]
    elif shape == "unique_pairs":
        i = rng.permutation(n).astype(np.int64)
        cols = [(i // 1000 - 77, None, False), (i % 1000 + 10**15, None, False)]
    elif shape == "one_hot_combination":
        a, b = rng.integers(0, 700, n, dtype=np.int64), rng.integers(-5, 5, n, dtype=np.int64)
        hot = rng.random(n) < 0.5
        a[hot], b[hot] = 123, 4
        cols = [(a, rng.random(n) < 0.01, False), (b, None, False)]
    elif shape == "doubles_one_binade":     # DOUBLE images that differ in their low bits only: packed
        cols = [(1.0 + rng.integers(0, 1000, n) * 2.0**-20, rng.random(n) < 0.1, True), (rng.integers(0, 100, n, dtype=np.int64), None, False)]
    elif shape == "all_null_column":
        cols = [(rng.integers(0, 9, n, dtype=np.int64), np.ones(n, dtype=bool), False), (rng.integers(0, 50, n, dtype=np.int64), rng.random(n) < 0.5, False)]
    else:
        cols = [(rng.integers(0, 3, n, dtype=np.int64), None, False) for _ in range(5)]
    keys_np = [(v, nl, None, dbl, False) for v, nl, dbl in cols]
    keys_dev = [(dev.to_dev(v), dev.nullbits_dev(nl) if nl is not None else None, None, Dbl if dbl else I, False) for v, nl, dbl in cols]
    first, cnt = dev.group_count_multi(keys_dev, n)
    ef, ec = orc.group_count_multi(keys_np, n)
    assert np.array_equal(_np(first).view(np.uint32).astype(np.int64), ef) and np.array_equal(_np(cnt), ec)


def _group_multi_numpy(cols, n):
    """first rows (ascending) and COUNTs of the combinations of `cols` = [(values, nulls or None)], vectorised: NULL = NULL, values by their bits"""
    parts = []
    for v, nl in cols:
        bits = np.ascontiguousarray(v).view(np.uint64).copy()
        if nl is not None:
            bits[nl] = 0
            parts.append(nl.astype(np.uint64))
        parts.append(bits)
    _, first, cnt = np.unique(np.stack(parts, axis=1), axis=0, return_index=True, return_counts=True)
    order = np.argsort(first, kind="stable")
    return first[order].astype(np.int64), cnt[order].astype(np.int64)


@pytest.mark.parametrize("shape", ["512x300", "nulls", "three_columns", "double_one_binade", "outlier_on_an_unsampled_row", "hot_combination", "17_bits", "26_bits",
                                   "desc_and_nulls", "unaligned_column", "16x50", "16x50_nulls", "14_bits", "four_flags", "16x50_outlier"])
@pytest.mark.parametrize("sample", ["2", "0"])
def test_group_count_multi_band_sort_reads_the_columns_itself(dev, shape, sample, monkeypatch):
    """GROUP BY over columns of 18 ... 25 bits together, 2^21 rows and more, no row-id vector: k_bg_band_sort<., true> builds the composite value from
    the columns (no composite column written).  Against numpy and against the form that writes the column (MDB_GROUP_MULTI_FUSED=0): NULLs, three
    columns, a DOUBLE column, a value outside the sampled ranges (the band sort reports it, the ranges are measured), a combination that holds
    half the rows (its region overflows: the other forms), 17 and 26 bits (not this form), a column that is not 16-byte aligned."""
    monkeypatch.setenv("MDB_SORT_RANGE_SAMPLE", sample)
    n = 2_300_001
    rng = np.random.default_rng(len(shape) + 11)
    I, Dbl = D.T_INT64, D.T_DOUBLE
    a = rng.integers(0, 512, n, dtype=np.int64)
    b = rng.integers(-150, 150, n, dtype=np.int64)
    cols, types, descs = [(a, None), (b, None)], [I, I], [False, False]
    if shape == "nulls":
        cols = [(a, rng.random(n) < 0.1), (b, rng.random(n) < 0.3)]
    elif shape == "three_columns":
        cols = [(a, None), (rng.integers(10**12, 10**12 + 40, n, dtype=np.int64), rng.random(n) < 0.05), (rng.integers(-3, 4, n, dtype=np.int64), None)]
        types, descs = [I, I, I], [False, False, False]
    elif shape == "double_one_binade":
        cols = [(1.0 + rng.integers(0, 1000, n) * 2.0**-52, None), (b, None)]
        types = [Dbl, I]
    elif shape == "outlier_on_an_unsampled_row":
        step = n // 2**17
        r = next(r for r in range(5000, 5400) if r != (r // step) * step + ((((r // step) * 0x9E3779B97F4A7C15) % 2**64) >> 32) % step)
        b[r] = 10**7
    elif shape == "hot_combination":
        hot = rng.random(n) < 0.5
        a[hot], b[hot] = 77, -3
    elif shape == "17_bits":
        cols = [(a, None), (rng.integers(0, 200, n, dtype=np.int64), None)]
    elif shape == "26_bits":
        cols = [(rng.integers(0, 2**13, n, dtype=np.int64), None), (rng.integers(0, 2**13, n, dtype=np.int64), None)]
    elif shape == "desc_and_nulls":
        cols = [(a, rng.random(n) < 0.02), (b, None)]
        descs = [True, True]
    elif shape in ("16x50", "16x50_nulls", "16x50_outlier"):       # at most 14 bits: per-workgroup LDS tables, the slot built from the columns
        c1, c2 = rng.integers(-8, 8, n, dtype=np.int64), rng.integers(10**9, 10**9 + 50, n, dtype=np.int64)
        if shape == "16x50_outlier":
            step = n // 2**17
            r = next(r for r in range(9000, 9400) if r != (r // step) * step + ((((r // step) * 0x9E3779B97F4A7C15) % 2**64) >> 32) % step)
            c2[r] = 7
        cols = [(c1, rng.random(n) < 0.2 if shape == "16x50_nulls" else None), (c2, rng.random(n) < 0.01 if shape == "16x50_nulls" else None)]
    elif shape == "14_bits":
        cols = [(a, None), (rng.integers(0, 32, n, dtype=np.int64), None)]
    elif shape == "four_flags":
        cols = [(rng.integers(0, 2, n, dtype=np.int64), rng.random(n) < 0.3) for _ in range(4)]
        types, descs = [I] * 4, [False, True, False, True]
    dev_cols = []
    for (v, nl), t in zip(cols, types):
        vd = dev.to_dev(v)
        if shape == "unaligned_column" and len(dev_cols) == 1:      # 8 bytes past a 16-byte boundary
            big = torch.empty(n + 1, dtype=vd.dtype, device=dev.device)
            big[1:] = vd
            vd = big[1:]
        dev_cols.append((vd, dev.nullbits_dev(nl) if nl is not None else None))
    keys_dev = [(vd, nd, None, t, d) for (vd, nd), t, d in zip(dev_cols, types, descs)]
    first, cnt = dev.group_count_multi(keys_dev, n)
    ef, ec = _group_multi_numpy(cols, n)
    assert np.array_equal(_np(first).view(np.uint32).astype(np.int64), ef) and np.array_equal(_np(cnt), ec)
    monkeypatch.setenv("MDB_GROUP_MULTI_FUSED", "0")
    first0, cnt0 = dev.group_count_multi(keys_dev, n)
    assert torch.equal(first0, first) and torch.equal(cnt0, cnt)
    monkeypatch.delenv("MDB_GROUP_MULTI_FUSED")
    sel = dev.distinct_sel(keys_dev, n)
    assert np.array_equal(_np(sel).view(np.uint32).astype(np.int64), ef)


def test_group_count_multi_large_property_unpinned(dev):
    """10^7 rows, key = (i mod 1000, i mod 7): 7000 groups (1000 and 7 are coprime), counts n/7000 +- 1, firsts = 0..6999."""
    n = 10_000_000
    i = torch.arange(n, dtype=torch.int64, device=dev.device)
    a, b = i % 1000, i % 7
    first, cnt = dev.group_count_multi([(a, None, None, D.T_INT64, False), (b, None, None, D.T_INT64, False)], n)
    assert first.numel() == 7000 and int(cnt.sum()) == n
    assert bool((first.long() == torch.arange(7000, device=dev.device)).all())
    assert int(cnt.min()) >= n // 7000 and int(cnt.max()) <= n // 7000 + 1


def test_partition_by_dest_4_byte_wire_format_and_key_range(dev):
    rng = np.random.default_rng(5)
    k = rng.integers(-2**31, 2**31, 300_001, dtype=np.int64)
    nl = rng.random(len(k)) < 0.02
    kd, nd = dev.to_dev(k), dev.nullbits_dev(nl)
    assert dev.key_range(kd, nd) == (int(k[~nl].min()), int(k[~nl].max()))
    assert dev.key_range(kd) == (int(k.min()), int(k.max()))
    out8, c8 = dev.partition_by_dest(kd, nd, 4)
    out4, c4, rid = dev.partition_by_dest(kd, nd, 4, with_rid=True, keys32=True)
    assert c4 == c8 and out4.dtype == torch.int32
    wide = dev.widen32(out4)
    r = _np(rid).view(np.uint32)
    assert np.array_equal(_np(wide), k[r])              # every 4-byte key widens back to its source row's key
    off = np.concatenate([[0], np.cumsum(c8)])
    for d in range(4):                                  # same multiset per destination as the 8-byte form
        assert np.array_equal(np.sort(_np(wide)[off[d]:off[d + 1]]), np.sort(_np(out8)[off[d]:off[d + 1]]))
    # a key outside the int32 range is reported, never truncated (a NULL row's stale value does not count)
    k2 = k.copy()
    k2[1234] = 2**31
    nl2 = nl.copy()
    nl2[1234] = False
    with pytest.raises(D.DeviceError, match="4-byte wire format"):
        dev.partition_by_dest(dev.to_dev(k2), dev.nullbits_dev(nl2), 4, keys32=True)
    nl2[1234] = True
    out_ok, c_ok = dev.partition_by_dest(dev.to_dev(k2), dev.nullbits_dev(nl2), 4, keys32=True)
    assert sum(c_ok) == int((~nl2).sum())


@pytest.mark.parametrize("shape", ["one_key", "two_keys", "hot_plus_unique", "many_hot_keys", "hot_left_only"])
def test_hot_keys(dev, shape):
    """Keys with 10^5..10^6 duplicates: the plain leaf kernel hands such leaves to the hot-key path (slices shared by all
    workgroups, merged through a table in global memory; more than 64 hot leaves: one workgroup per leaf).  Exact
    result against the C hash oracle, for the join and for the plain GROUP BY."""
    from oracle import cpu
    rng = np.random.default_rng(len(shape))
    n_l, n_r = 3_000_000, 1_000_000
    if shape == "one_key":
        kl, kr = np.full(n_l, 7, dtype=np.int64), np.full(n_r, 7, dtype=np.int64)
    elif shape == "two_keys":
        kl, kr = rng.integers(0, 2, n_l), rng.integers(0, 2, n_r)
    elif shape == "hot_plus_unique":
        kl = np.where(rng.random(n_l) < 0.5, 0, np.arange(n_l) + 10)
        kr = np.where(rng.random(n_r) < 0.3, 0, rng.integers(10, n_l, n_r))
    elif shape == "many_hot_keys":          # 150 keys x ~20000 rows each: more hot leaves than the shared path takes
        kl, kr = rng.integers(0, 150, n_l), rng.integers(0, 150, n_r)
    else:                                   # only the left side is hot; the right side has the key three times
        kl = np.where(rng.random(n_l) < 0.4, 5, np.arange(n_l) + 10)
        kr = np.concatenate([np.array([5, 5, 5]), rng.integers(10, n_l, n_r - 3)])
    kl, kr = kl.astype(np.int64), kr.astype(np.int64)
    nl = (rng.random(n_l) < 0.01) if shape == "hot_plus_unique" else None
    ek, ec, ef, ej = cpu.hash_join_group_count(kl, nl, kr, None, 8)
    k, c, f, j = dev.join_group_count(dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr), None)
    assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec)
    assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef)
    first, cnt = dev.group_count(dev.to_dev(kl), dev.nullbits_dev(nl))
    e_first, e_cnt = orc.group_count(kl, nl)
    assert np.array_equal(_np(first).view(np.uint32).astype(np.int64), e_first) and np.array_equal(_np(cnt), e_cnt)


def test_combine_counts(dev):
    rng = np.random.default_rng(9)
    g1, n = 5000, 3000
    cnt1 = rng.integers(1, 50, g1, dtype=np.int64)
    first1 = rng.permutation(100_000)[:g1].astype(np.uint32)
    idx = np.sort(rng.choice(g1, n, replace=False)).astype(np.uint32)
    cnt2 = rng.integers(1, 9, n, dtype=np.int64)
    out, outf, tot = dev.combine_counts(dev.to_dev(cnt1), dev.to_dev(first1), dev.to_dev(idx), dev.to_dev(cnt2))
    assert np.array_equal(_np(out), cnt1[idx] * cnt2) and np.array_equal(_np(outf).view(np.uint32), first1[idx])
    assert tot == int((cnt1[idx] * cnt2).sum())
    out, outf, tot = dev.combine_counts(dev.to_dev(cnt1), None, dev.to_dev(idx), dev.to_dev(cnt2))
    assert np.array_equal(_np(outf).view(np.uint32), idx)


# ---- narrow form of the join (32-bit hashes, row id inside the word): same results as the 64-bit form ----

@pytest.fixture
def narrow_mode(dev):
    def set_mode(m):
        dev.set_narrow_keys(m)
    yield set_mode
    dev.set_narrow_keys(1)


def _jgc_check(dev, kl, nl, kr, nr):
    ek, ec, ef, ej = orc.join_group_count(kl, nl, kr, nr)
    k, c, f, j = dev.join_group_count(dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr), dev.nullbits_dev(nr))
    assert j == ej
    assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef)
    assert np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec)
    first, cnt = dev.group_count(dev.to_dev(kl), dev.nullbits_dev(nl))
    e_first, e_cnt = orc.group_count(kl, nl)
    assert np.array_equal(_np(first).view(np.uint32).astype(np.int64), e_first) and np.array_equal(_np(cnt), e_cnt)


@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("n_l,n_r,domain,null_frac,lo", [c for c in CASES_JGC if c[0] != 300_000])
def test_narrow_and_wide_forms_agree_with_the_oracle(dev, narrow_mode, mode, n_l, n_r, domain, null_frac, lo):
    narrow_mode(mode)
    rng = np.random.default_rng(n_l * 17 + n_r + mode)
    kl, nl = _mk(rng, n_l, domain, null_frac, lo)
    kr, nr = _mk(rng, n_r, domain, null_frac, lo)
    _jgc_check(dev, kl, nl, kr, nr)


def test_narrow_form_covers_the_whole_int32_range(dev, narrow_mode):
    narrow_mode(2)
    rng = np.random.default_rng(77)
    edge = np.array([-2**31, 2**31 - 1, 0, -1, 1], dtype=np.int64)
    kl = np.concatenate([edge, rng.integers(-2**31, 2**31, 50_000, dtype=np.int64), edge])
    kr = np.concatenate([rng.choice(kl, 80_000), edge, edge])
    _jgc_check(dev, kl, None, kr, None)


@pytest.mark.parametrize("where", ["left", "right", "both", "null_only"])
@pytest.mark.parametrize("mode,n", [(2, 20_000), (1, 1_300_000)])
def test_narrow_form_is_abandoned_when_a_key_is_outside_the_range(dev, narrow_mode, where, mode, n):
    """One key just outside [-2^31, 2^31) hidden among n others (the sample of mode 1 does not see it): the level-0
    check raises the flag and the operator redoes its work with 64-bit hashes.  A wide value under a NULL bit is no key."""
    narrow_mode(mode)
    rng = np.random.default_rng(n + len(where))
    kl = rng.integers(-1000, n // 3, n, dtype=np.int64)
    kr = rng.integers(-1000, n // 3, n + 17, dtype=np.int64)
    nl = np.zeros(n, dtype=bool)
    nr = np.zeros(n + 17, dtype=bool)
    wide = [2**31, -2**31 - 1, 2**40 + 5, (2**31 - 1) + 2**32]      # the last one equals 2^31-1 in its low 32 bits
    if where in ("left", "both"):
        kl[[n // 7 * 2 + 1, n - 1]] = wide[:2]
        kl[5] = wide[3]
        kl[6] = 2**31 - 1
    if where in ("right", "both"):
        kr[[3, n // 2 + 3]] = wide[2:]
        kr[4] = 2**31 - 1
        kl[7] = wide[3]
    if where == "null_only":
        kl[11] = wide[2]
        nl[11] = True
        kr[13] = wide[0]
        nr[13] = True
    _jgc_check(dev, kl, nl, kr, nr)


def test_narrow_form_large_property(dev, narrow_mode):
    """2*10^7 x 2*10^7 rows, default mode (sampled): both forms deliver identical columns."""
    n = 20_000_000
    kl = dev.gen_keys(n, 0, n, 42, 0)
    kr = dev.gen_keys(n, 0, n, 43, n // 16)
    out = {}
    for mode in (0, 1):
        narrow_mode(mode)
        k, c, f, j = dev.join_group_count(kl, None, kr, None)
        out[mode] = (k.clone(), c.clone(), f.clone(), j)
    for a, b in zip(out[0][:3], out[1][:3]):
        assert torch.equal(a, b)
    assert out[0][3] == out[1][3] == n


@pytest.mark.parametrize("case", ["narrow_both", "right_wide", "left_wide", "right_all_null", "right_tiny", "right_empty_leaves"])
def test_narrow_form_split_begin_finish_and_ragged_right_sides(dev, narrow_mode, case):
    """The split operator decides the form from the left table alone at begin(); a right table with a key outside the
    int32 range (seen only while it is partitioned at finish()) makes the whole operator run again wide.  Right sides
    that leave most 4-byte leaf regions empty or hold NULLs only go through the same kernels."""
    narrow_mode(2)
    rng = np.random.default_rng(len(case))
    n_l, n_r = 1_100_000, 700_000
    kl = rng.integers(-400_000, 400_000, n_l, dtype=np.int64)
    kr = rng.integers(-400_000, 400_000, n_r, dtype=np.int64)
    nl = rng.random(n_l) < 0.01
    nr = None
    if case == "right_wide":
        kr[n_r // 3] = 2**33 + 7
    elif case == "left_wide":
        kl[n_l // 5] = -2**35
    elif case == "right_all_null":
        nr = np.ones(n_r, dtype=bool)
    elif case == "right_tiny":
        kr, n_r = kr[:37].copy(), 37
    elif case == "right_empty_leaves":
        kr = rng.integers(100, 164, n_r, dtype=np.int64)          # 64 distinct keys: hot leaves, the rest empty
    ek, ec, ef, ej = orc.join_group_count(kl, nl, kr, nr)
    dl, dnl, dr, dnr = dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr), dev.nullbits_dev(nr)
    for split in (False, True):
        if split:
            dev.join_group_count_begin(dl, dnl, n_r + 100)
            k, c, f, j = dev.join_group_count_finish(dr, dnr)
        else:
            k, c, f, j = dev.join_group_count(dl, dnl, dr, dnr)
        assert j == ej
        assert np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec)
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef)
    if case == "narrow_both":
        assert dev.last_join_narrow()
    if case in ("right_wide", "left_wide"):
        assert not dev.last_join_narrow()


def test_narrow_form_decision_is_remembered_per_column_but_every_key_is_still_checked(dev, narrow_mode):
    """Second query over the same key columns skips the sampling kernel; when the column's contents changed in place
    (a key outside the int32 range now), the level-0 check still sends the operator to the wide form - and the
    remembered decision flips, so the third query goes wide directly."""
    narrow_mode(1)
    rng = np.random.default_rng(5)
    n = 1_200_000
    kl = rng.integers(0, 300_000, n, dtype=np.int64)
    kr = rng.integers(0, 300_000, n, dtype=np.int64)
    dl, dr = dev.to_dev(kl), dev.to_dev(kr)
    for round_ in range(4):
        if round_ == 2:
            # (in the RIGHT column: a left key outside the right table's range is dropped by min-max pruning - exact, and no reason
            # to leave the narrow form - whenever the operator partitions the right table first)
            kr[n // 2] = 2**40
            dr[n // 2] = 2**40
        ek, ec, ef, ej = orc.join_group_count(kl, None, kr, None)
        k, c, f, j = dev.join_group_count(dl, None, dr, None)
        assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec)
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef)
        assert dev.last_join_narrow() == (round_ < 2)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("n_l,n_r,dom", [(1, 1, 1), (63, 130, 20), (5000, 4000, 900), (400_000, 500_000, 150_000),
                                         (1_300_000, 1_100_001, 2_000_000), (3_000_000, 700_000, 40_000)])
def test_join_group_count_over_int32_key_columns(dev, narrow_mode, mode, n_l, n_r, dom):
    """The operator's int32 entry points (keys that crossed xGMI in the 4-byte wire format are consumed as they
    arrive): same results as the int64 operator on the widened columns, one call and split, whole int32 range."""
    narrow_mode(mode)
    rng = np.random.default_rng(n_l + 3 * n_r + mode)
    lo = int(rng.integers(-2**31, 2**31 - dom))
    kl = rng.integers(lo, lo + dom, n_l, dtype=np.int64)
    kr = rng.integers(lo, lo + dom, n_r, dtype=np.int64)
    kl[0], kr[0] = -2**31, 2**31 - 1
    if n_r > 1:
        kr[1] = -2**31
    ek, ec, ef, ej = orc.join_group_count(kl, None, kr, None)
    dl = torch.from_numpy(kl.astype(np.int32)).to(dev.device)
    dr = torch.from_numpy(kr.astype(np.int32)).to(dev.device)
    for split in (False, True):
        if split:
            dev.join_group_count_begin(dl, None, n_r + 5)
            k, c, f, j = dev.join_group_count_finish(dr, None)
        else:
            k, c, f, j = dev.join_group_count_i32(dl, dr)
        assert j == ej and k.dtype == torch.int64
        assert np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec)
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef)


@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("n_l,n_r,shape", [(7, 5, "unique"), (5000, 3000, "unique"), (600_000, 500_000, "unique"), (1_300_000, 1_200_000, "unique"),
                                           (40_000, 30_000, "dups"), (900_000, 450_000, "wide_right"), (900_000, 450_000, "wide_left_null"),
                                           (300_000, 200_000, "edges")])
def test_join_pairs_narrow_and_wide_forms(dev, narrow_mode, mode, n_l, n_r, shape):
    """The materialising join with unique right keys in both forms (narrow: both sides travel as hash32 | row id words):
    same (l, r) pairs in left-row order as the oracle; a key outside the int32 range sends the narrow attempt back to
    64-bit hashes, a wide value under a NULL bit does not; duplicates on the right go on to the general path."""
    narrow_mode(mode)
    rng = np.random.default_rng(n_l + n_r + mode)
    kr = rng.permutation(3 * n_r)[:n_r].astype(np.int64) - n_r                 # unique right keys, some negative
    kl = rng.integers(-n_r, 2 * n_r, n_l, dtype=np.int64)
    nl = rng.random(n_l) < 0.02
    nr = None
    if shape == "dups":
        kr[: n_r // 3] = kr[n_r // 3: 2 * (n_r // 3)]
    elif shape == "wide_right":
        kr[n_r // 2] = 2**32 + int(kr[n_r // 2 + 1])                              # equal to another key in its low 32 bits
        kl[17] = kr[n_r // 2]
    elif shape == "wide_left_null":
        kl[33] = 2**45
        nl[33] = True
    elif shape == "edges":
        kr[:2] = [-2**31, 2**31 - 1]
        kl[:4] = [2**31 - 1, -2**31, -2**31, 0]
        nl[:4] = False
    el, er = orc.join_pairs(kl, nl, kr, nr)
    l, r = dev.join_pairs(dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr), dev.nullbits_dev(nr))
    assert l.numel() == len(el)
    assert np.array_equal(_np(l).astype(np.int64), el) and np.array_equal(_np(r).astype(np.int64), er)


@pytest.mark.parametrize("i32", [False, True])
def test_split_operator_with_a_right_table_larger_than_announced(dev, i32):
    """A skewed exchange can hand a GPU more right rows than begin() was told: finish() still delivers the result."""
    rng = np.random.default_rng(11)
    n_l, n_r = 700_000, 2_500_000
    kl = rng.integers(0, 50_000, n_l, dtype=np.int64)
    kr = np.where(rng.random(n_r) < 0.5, 7, rng.integers(0, 50_000, n_r)).astype(np.int64)     # half the rows carry one key
    ek, ec, ef, ej = orc.join_group_count(kl, None, kr, None)
    if i32:
        dl, dr = torch.from_numpy(kl.astype(np.int32)).to(dev.device), torch.from_numpy(kr.astype(np.int32)).to(dev.device)
    else:
        dl, dr = dev.to_dev(kl), dev.to_dev(kr)
    dev.join_group_count_begin(dl, None, 900_000)
    k, c, f, j = dev.join_group_count_finish(dr, None)
    assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec)
    assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef)


@pytest.mark.parametrize("kind,desc,nf,with_rid", [
    ("neg", False, 0.0, False), ("neg", True, 0.15, False), ("small", False, 0.0, True), ("small", True, 0.3, False),
    ("double", False, 0.0, False), ("double", True, 0.2, True), ("const", False, 0.0, False), ("const", True, 0.5, False),
    ("full", False, 0.0, False), ("small", False, 1.0, False),
])
@pytest.mark.parametrize("n", [262_144, 300_001, 1_500_000])
def test_sort_perm_single_key_packed_path_unpinned(dev, kind, desc, nf, with_rid, n):
    """Single ORDER BY column from 2^18 rows on: value range and stream position share one word, partitioned by its top
    bits and finished per leaf in LDS.  Against the numpy oracle: ascending / descending, NULLs first / last, DOUBLE
    (incl. -0.0 / 0.0), all-equal and all-NULL columns, through a row-id vector; 'full' (64-bit range) does not fit
    a word and takes the general path."""
    _sort_case(dev, np.random.default_rng(n + len(kind)), n, [(kind, desc, nf)], with_rid=with_rid)


@pytest.mark.parametrize("shape", ["bunched_high", "bunched_low", "two_values", "sorted", "reversed", "geometric"])
def test_sort_perm_single_key_uneven_value_distributions_unpinned(dev, shape):
    """Value distributions the fixed-capacity leaves of the packed path cannot hold (the top bits are the values
    themselves) are detected on the device and sorted by the general path: same permutation either way."""
    n = 700_000
    rng = np.random.default_rng(len(shape))
    if shape == "bunched_high":
        v = np.where(rng.random(n) < 0.9, 2**40 + rng.integers(0, 50, n), rng.integers(0, 2**41, n))
    elif shape == "bunched_low":
        v = rng.integers(0, 4, n) * 2**30                      # four values, far apart: low bits of the word all equal per value
    elif shape == "two_values":
        v = np.where(rng.random(n) < 0.5, -5, 10**12)
    elif shape == "sorted":
        v = np.arange(n) * 3 - 1000
    elif shape == "reversed":
        v = (n - np.arange(n)) * 7
    else:
        v = np.floor(np.exp(rng.uniform(0, 40, n))).astype(np.int64)
    v = v.astype(np.int64)
    nulls = rng.random(n) < 0.01
    for desc in (False, True):
        got = _np(dev.sort_perm([(dev.to_dev(v), dev.nullbits_dev(nulls), None, D.T_INT64, desc)], n)).view(np.uint32)
        want = orc.sort_perm([(v, nulls, None, False, desc)], n)
        assert np.array_equal(got, want)


def test_sort_perm_packed_path_large_property_unpinned(dev):
    """10^8 rows, one key: a permutation, keys non-decreasing, ties in stream order."""
    n = 100_000_000
    for modulus in (0, 5000):
        a = dev.gen_keys(n, 0, n, 9, modulus)
        perm = dev.sort_perm([(a, None, None, D.T_INT64, False)], n).to(torch.int64)
        s = a[perm]
        assert bool((s[1:] >= s[:-1]).all())
        tie = s[1:] == s[:-1]
        assert bool((perm[1:][tie] > perm[:-1][tie]).all())
        assert int(perm.sum()) == n * (n - 1) // 2 and int(perm.min()) == 0 and int(perm.max()) == n - 1
        del perm, s, tie
        torch.cuda.empty_cache()


@pytest.mark.parametrize("specs", [
    [("small", False, 0.0), ("neg", True, 0.0)], [("small", True, 0.3), ("neg", False, 0.2), ("small", False, 0.0)],
    [("neg", False, 0.0), ("const", True, 0.4), ("neg", True, 0.0), ("small", False, 0.1)],
    [("small", False, 0.0), ("neg", True, 0.0), ("small", True, 0.0), ("neg", False, 0.0), ("small", False, 0.0)],     # 5 keys: general path
    [("small", False, 0.0), ("double", True, 0.1)],                                                                       # DOUBLE range: general path
    [("neg", True, 1.0), ("small", False, 0.0)],
], ids=lambda s: "+".join(f"{k}{'D' if d else 'A'}{int(nf * 10)}" for k, d, nf in s))
def test_sort_perm_multi_key_packed_path_unpinned(dev, specs):
    """Several ORDER BY columns whose ranges fit one word together (composite word, most significant column first)."""
    rng = np.random.default_rng(len(specs) * 13 + 5)
    _sort_case(dev, rng, 400_003, specs)
    _sort_case(dev, rng, 300_000, specs, with_rid=True)


@pytest.mark.parametrize("shape", ["even", "outlier_above_unsampled", "outlier_below_unsampled", "nulls_in_every_sampled_row", "widened_does_not_fit",
                                   "at_int64_min", "at_int64_max"])
@pytest.mark.parametrize("mode", ["2", "0"])
def test_packed_words_with_ranges_from_a_sample(dev, shape, mode, monkeypatch):
    """The packed word's column ranges come from a sample of 2^17 rows first (MDB_SORT_RANGE_SAMPLE=2: from 2^18 rows on; by default from 2^22),
    widened by a 1024th; the packing kernel checks every row and the ranges are measured when one lies outside: an extreme value on a row
    the sample skips, a column that is NULL wherever the sample looks, ranges that fit the word only unwidened, values at either end of
    int64 (the widening must not wrap) - ORDER BY, GROUP BY over two columns and DISTINCT against the numpy oracle, sample on and off."""
    monkeypatch.setenv("MDB_SORT_RANGE_SAMPLE", mode)
    n = 600_001
    step = n // 2**17

    def sampled(r):     # k_sort_ranges: the row looked at in the k-th stretch of `step` rows
        k = r // step
        return k < 2**17 and r == k * step + (((k * 0x9E3779B97F4A7C15) % 2**64) >> 32) % step
    unsampled = [r for r in range(4000, 4400) if not sampled(r)]
    rng = np.random.default_rng(len(shape) + 3)
    a = rng.integers(-50, 50, n, dtype=np.int64)
    b = rng.integers(10**12, 10**12 + 300, n, dtype=np.int64)
    na = None
    if shape == "outlier_above_unsampled":
        a[unsampled[0]] = 10**6
    elif shape == "outlier_below_unsampled":
        b[unsampled[1]] = -(10**9)
    elif shape == "nulls_in_every_sampled_row":
        na = np.ones(n, dtype=bool)
        na[unsampled[:50]] = False
    elif shape == "widened_does_not_fit":       # 22 + 22 value bits + 20 position bits = 64: one more bit per column does not fit
        a = rng.integers(0, 2**22, n, dtype=np.int64)
        b = rng.integers(0, 2**22, n, dtype=np.int64)
        a[:2] = [0, 2**22 - 1]
        b[:2] = [0, 2**22 - 1]
    elif shape == "at_int64_min":
        a = np.iinfo(np.int64).min + rng.integers(0, 100, n, dtype=np.int64)
    elif shape == "at_int64_max":
        a = np.iinfo(np.int64).max - rng.integers(0, 100, n, dtype=np.int64)
    ad, bd, nad = dev.to_dev(a), dev.to_dev(b), dev.nullbits_dev(na) if na is not None else None
    for desc in (False, True):
        got = _np(dev.sort_perm([(ad, nad, None, D.T_INT64, desc), (bd, None, None, D.T_INT64, not desc)], n)).view(np.uint32)
        assert np.array_equal(got, orc.sort_perm([(a, na, None, False, desc), (b, None, None, False, not desc)], n))
    keys_np, keys_dev = [(a, na, None, False, False), (b, None, None, False, False)], [(ad, nad, None, D.T_INT64, False), (bd, None, None, D.T_INT64, False)]
    if shape != "widened_does_not_fit":          # (the python oracle walks every row; 4 x 10^12 combinations: 600 001 groups)
        first, cnt = dev.group_count_multi(keys_dev, n)
        ef, ec = orc.group_count_multi(keys_np, n)
        assert np.array_equal(_np(first).view(np.uint32).astype(np.int64), ef) and np.array_equal(_np(cnt), ec)
        assert np.array_equal(_np(dev.distinct_sel(keys_dev, n)).view(np.uint32), orc.distinct_sel(keys_np, n))


@pytest.mark.parametrize("shape", ["high_window", "negative_window", "outlier_above", "outlier_below", "too_wide", "int64_extremes"])
def test_narrow_form_window_anywhere_in_the_int64_range(dev, narrow_mode, shape):
    """Keys inside a 2^32-wide window far from zero (surrogate keys from 10^12 on, negative timestamps) take the narrow
    form relative to the centre of the sampled window; a key outside the window - hidden from the sample - sends the
    operator back to 64-bit hashes; sampled spans of 2^31 or more never try."""
    narrow_mode(1)
    rng = np.random.default_rng(len(shape) + 100)
    n_l, n_r = 1_400_000, 1_100_000
    off = {"high_window": 10**12, "negative_window": -(2**61), "outlier_above": 7 * 10**15, "outlier_below": 10**10,
           "too_wide": 5 * 10**9, "int64_extremes": 0}[shape]
    span = 2**33 if shape == "too_wide" else 900_000
    kl = off + rng.integers(0, span, n_l, dtype=np.int64)
    kr = off + rng.integers(0, span, n_r, dtype=np.int64)
    if shape == "outlier_above":
        kl[n_l // 2 + 1] = off + 2**32 + 17          # same low 32 bits as off + 17
        kr[5] = off + 17
        kl[7] = off + 17
    elif shape == "outlier_below":
        kr[n_r // 2 + 1] = off - 2**31 - 10
    elif shape == "int64_extremes":
        kl[:2] = [np.iinfo(np.int64).min, np.iinfo(np.int64).max]
        kr[:2] = [np.iinfo(np.int64).max, np.iinfo(np.int64).min]
    nl = rng.random(n_l) < 0.01
    ek, ec, ef, ej = orc.join_group_count(kl, nl, kr, None)
    dl, dnl, dr = dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr)
    k, c, f, j = dev.join_group_count(dl, dnl, dr, None)
    assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec)
    assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef)
    assert dev.last_join_narrow() == (shape in ("high_window", "negative_window"))
    first, cnt = dev.group_count(dl, dnl)
    e_first, e_cnt = orc.group_count(kl, nl)
    assert np.array_equal(_np(first).view(np.uint32).astype(np.int64), e_first) and np.array_equal(_np(cnt), e_cnt)
    # materialising join on unique right keys of the same window
    kr_u = off + rng.permutation(span if span < 2**31 else 3_000_000)[:n_r].astype(np.int64)
    el, er = orc.join_pairs(kl, nl, kr_u, None)
    l, r = dev.join_pairs(dl, dnl, dev.to_dev(kr_u), None)
    assert np.array_equal(_np(l).astype(np.int64), el) and np.array_equal(_np(r).astype(np.int64), er)


@pytest.mark.parametrize("shape", ["dense_perm", "dup16", "offset_1e12", "negative", "nulls", "n_to_m", "rem4", "rem12", "sparse_not_direct",
                                   "outlier_left", "outlier_right", "hot_key", "left_small", "right_small"])
def test_compact_narrow_form_direct_address_leaves(dev, narrow_mode, shape):
    """Keys that span fewer than 2^k values hash into k bits (mdb_mixk of key - window base, verified per key); what the
    radix partition leaves of them indexes the leaf tables directly (k_leaf_direct).  Same groups, counts, first rows and
    order as the oracle for dense and duplicated keys, windows anywhere in the int64 range, NULLs, N:M duplicates, the
    smallest and largest table sizes, a hot key (left to the hot-key path), and a key outside the sampled window - hidden
    from the sample - which sends the operator to the plain narrow form."""
    narrow_mode(1)
    rng = np.random.default_rng(len(shape) * 7 + 1)
    n_l, n_r = 1_500_000, 1_300_000
    span, off, expect = 1_500_000, 0, 2
    if shape == "offset_1e12":
        off = 10**12
    elif shape == "negative":
        off = -(2**40)
    elif shape == "rem4":
        span = 12_000		# 14-bit window over 2^10 leaves
    elif shape == "rem12":
        span = 3_500_000	# 22-bit window
    elif shape == "sparse_not_direct":
        span, expect = 2**27, 1
    elif shape == "left_small":
        n_l = 500_000
    elif shape == "right_small":
        n_r = 5000
    if shape == "dense_perm":
        kl = off + rng.permutation(span)[:n_l].astype(np.int64)
        kr = off + rng.permutation(span)[:n_r].astype(np.int64)
    elif shape == "dup16":
        kl = off + rng.permutation(span)[:n_l].astype(np.int64)
        kr = off + (rng.permutation(span)[:n_r] % (span // 16)).astype(np.int64)
    else:
        kl = off + rng.integers(0, span, n_l, dtype=np.int64)
        kr = off + rng.integers(0, span, n_r, dtype=np.int64)
    nl = nr = None
    if shape == "nulls":
        nl, nr = rng.random(n_l) < 0.05, rng.random(n_r) < 0.2
        kl[nl] = rng.integers(-2**62, 2**62, int(nl.sum()))	# whatever lies under a NULL bit is no key
    if shape == "outlier_left":
        kl[n_l // 3 + 1] = off + 5 * span			# beyond the padded, rounded-up window of the sample, inside the 2^32 one
        kr[7] = kl[n_l // 3 + 1]
        expect = 1
    if shape == "outlier_right":
        kr[n_r - 2] = off - 4 * span
        kl[11] = kr[n_r - 2]
        expect = 1
    if shape == "hot_key":
        kr[100_000:500_000] = off + 12345
        kl[50_000:50_040] = off + 12345
        expect = 0		# the hot leaf outgrows its fixed-capacity region: exact layout, 64-bit hashes, hot-key path
    ek, ec, ef, ej = orc.join_group_count(kl, nl, kr, nr)
    dl, dr, dnl, dnr = dev.to_dev(kl), dev.to_dev(kr), dev.nullbits_dev(nl), dev.nullbits_dev(nr)
    for round_ in range(2):		# the second call runs on the remembered verdict
        k, c, f, j = dev.join_group_count(dl, dnl, dr, dnr)
        assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec)
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef)
        assert dev.last_join_form() == expect, (shape, round_, dev.last_join_form())
    first, cnt = dev.group_count(dl, dnl)
    e_first, e_cnt = orc.group_count(kl, nl)
    assert np.array_equal(_np(first).view(np.uint32).astype(np.int64), e_first) and np.array_equal(_np(cnt), e_cnt)
    # right side grouped on its own (duplicates / the hot key on the build side of a plain GROUP BY)
    first, cnt = dev.group_count(dr, dnr)
    e_first, e_cnt = orc.group_count(kr, nr)
    assert np.array_equal(_np(first).view(np.uint32).astype(np.int64), e_first) and np.array_equal(_np(cnt), e_cnt)


def test_compact_and_hashed_leaf_kernels_agree_at_scale(dev, narrow_mode):
    """2 * 10^7 x 2 * 10^7 rows of the benchmark's generator: the compact form (direct-address leaves), the plain narrow form
    and the wide form deliver identical columns."""
    n = 20_000_000
    kl = dev.gen_keys(n, 0, n, 42, 0)
    out = {}
    for variant, mod in (("D", n // 16), ("U", 0)):
        kr = dev.gen_keys(n, 0, n, 43, mod)
        for mode in (0, 2, 1):
            narrow_mode(mode)
            k, c, f, j = dev.join_group_count(kl, None, kr, None)
            assert dev.last_join_form() == {0: 0, 2: 1, 1: 2}[mode]
            out[mode] = (k.clone(), c.clone(), f.clone(), j)
        for m in (2, 1):
            for a, b in zip(out[0][:3], out[m][:3]):
                assert torch.equal(a, b)
            assert out[0][3] == out[m][3] == n


def test_join_group_count_8e8_rows_per_table_on_one_gpu(dev, narrow_mode):
    """Beyond the hashed leaf kernel's table capacity (7 * 10^8 build rows at 2^18 leaves): the direct-address leaves of the
    compact narrow form have no table to overflow.  8 * 10^8 x 8 * 10^8 rows of the benchmark's generator (variant D),
    checked through size-independent properties: groups, joined rows, every COUNT = 16, keys unique, first rows ascending."""
    narrow_mode(1)
    n = 800_000_000
    kl = dev.gen_keys(n, 0, n, 42, 0)
    kr = dev.gen_keys(n, 0, n, 43, n // 16)
    k, c, f, j = dev.join_group_count(kl, None, kr, None)
    assert dev.last_join_form() == 2
    assert k.numel() == n // 16 and j == n and int(c.sum()) == n and int(c.min()) == 16 and int(c.max()) == 16
    assert int(torch.unique(k).numel()) == k.numel()
    fi = f.to(torch.int64) & 0xFFFFFFFF
    assert bool((fi[1:] > fi[:-1]).all())
    assert bool((kl[fi] == k).all())          # a group's key is the key of its first left row
    del kl, kr, k, c, f, fi
    torch.cuda.empty_cache()


@pytest.mark.parametrize("seed", range(int(os.environ.get("MDB_FUZZ_SEEDS", "24"))))	# (MDB_FUZZ_SEEDS=300: the soak run of tests/soak/README.md)
def test_join_group_count_random_shapes_every_form_and_pruning_path(dev, narrow_mode, seed):
    """Randomised shapes through whatever form and pruning path the operator picks for them (compact / plain narrow / wide,
    min-max pruning, bitmap, keyed records, their retries): table sizes from 3 * 10^5 to 3 * 10^6 rows, key ranges from dense
    to 2^40-wide, the right table anywhere inside, beside or across the left table's range, duplicate factors, NULL fractions,
    windows anywhere in the int64 range - always the oracle's groups, counts, first rows and order.  (224 seeds ran clean when
    the paths were written; 24 stay in the suite.)"""
    narrow_mode(1)
    rng = np.random.default_rng(1000 + seed)
    n_l = int(rng.integers(300_000, 3_000_000))
    n_r = int(rng.integers(1, 3_000_000)) if rng.random() < 0.8 else int(rng.integers(1, 5000))
    span_l = int(n_l * float(rng.choice([1.0, 1.5, 4.0, 40.0, 3000.0]))) if rng.random() < 0.85 else 2**40
    off = int(rng.choice([0, 10**12, -(2**50), 2**31 - n_l // 2, -(2**31)]))
    kl = rng.integers(0, span_l, n_l, dtype=np.int64) if rng.random() < 0.5 else rng.permutation(max(span_l, n_l))[:n_l].astype(np.int64) if span_l < 2**27 else rng.integers(0, span_l, n_l, dtype=np.int64)
    frac = float(rng.choice([1.0, 0.5, 0.2, 1 / 16, 0.01]))
    start = int(rng.integers(0, max(1, int(span_l * (1.3 - frac)))))		# may stick out of the left table's range
    width = max(1, int(span_l * frac))
    dup = int(rng.choice([1, 1, 3, 16]))
    kr = start + (rng.integers(0, width, n_r, dtype=np.int64) // dup) * dup
    nl = (rng.random(n_l) < 0.03) if rng.random() < 0.3 else None
    nr = (rng.random(n_r) < 0.1) if rng.random() < 0.3 else None
    kl, kr = kl + off, kr + off
    ek, ec, ef, ej = orc.join_group_count(kl, nl, kr, nr)
    dl, dr, dnl, dnr = dev.to_dev(kl), dev.to_dev(kr), dev.nullbits_dev(nl), dev.nullbits_dev(nr)
    for _ in range(2):		# (the second call runs on what the first one learned about the columns)
        k, c, f, j = dev.join_group_count(dl, dnl, dr, dnr)
        info = (seed, n_l, n_r, span_l, off, frac, start, dup, dev.last_join_form(), dev.last_join_filter())
        assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec), info
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef), info


@pytest.mark.parametrize("seed", range(12))
def test_join_group_count_random_shapes_with_tiles_of_twice_the_rows(dev, narrow_mode, seed):
    """Tables of 2^25 rows and more go through the histogram-free first levels in tiles of 2 x 4096 rows (k_part_scatter<..._t2>): the same
    shape fuzz with that form switched on for the test's small tables (MDB_TILE2_MIN) - partial last tiles, NULLs, pruned left tables,
    the ordering sort's first level."""
    os.environ["MDB_TILE2_MIN"] = "1"
    try:
        test_join_group_count_random_shapes_every_form_and_pruning_path(dev, narrow_mode, 100 + seed)
    finally:
        del os.environ["MDB_TILE2_MIN"]


@pytest.mark.parametrize("case", ["variant_d_forced_wide", "hashes_2e62", "right_sticks_out", "disjoint", "nulls_dups", "not_prunable"])
def test_min_max_pruning_in_the_64_bit_form(dev, narrow_mode, case):
    """Keys that fit no 2^32 window (hashes, snowflake ids) - or the narrow forms switched off - still prune the left table by the right
    table's key range: the right table's first level records its smallest and largest KEY, the left table's drops the rows outside.
    Same groups, counts, first rows and order as the oracle; the operator says which form ran and whether it pruned."""
    rng = np.random.default_rng(len(case) * 7 + 1)
    nl = nr = None
    n = 1_600_000
    if case == "variant_d_forced_wide":
        narrow_mode(0)
        kl = rng.permutation(n).astype(np.int64)
        kr = rng.integers(0, n // 16, n, dtype=np.int64)
        expect_pruned = True
    else:
        narrow_mode(1)
        pool = np.unique(rng.integers(-2**62, 2**62, n, dtype=np.int64))		# sorted: a slice of it is a key RANGE
        kl = rng.permutation(pool)
        if case == "hashes_2e62":
            kr = pool[len(pool) // 3: len(pool) // 3 + len(pool) // 20][rng.integers(0, len(pool) // 20, n // 2)]
            expect_pruned = True
        elif case == "right_sticks_out":
            kr = np.concatenate([pool[-(len(pool) // 10):][rng.integers(0, len(pool) // 10, n // 3)], pool[-1] + 1 + rng.integers(0, 2**40, 1000)])
            expect_pruned = True
        elif case == "disjoint":
            kr = pool[-1] + 5 + rng.integers(0, 2**50, n // 4)
            expect_pruned = True
        elif case == "nulls_dups":
            kr = pool[: len(pool) // 8][rng.integers(0, len(pool) // 8, n)]
            nl, nr = rng.random(len(kl)) < 0.03, rng.random(len(kr)) < 0.1
            expect_pruned = True
        else:
            kr = pool[rng.integers(0, len(pool), n // 2)]		# spread over the whole range: nothing to prune
            expect_pruned = False
    kl, kr = kl.astype(np.int64), kr.astype(np.int64)
    ek, ec, ef, ej = orc.join_group_count(kl, nl, kr, nr)
    dl, dr, dnl, dnr = dev.to_dev(kl), dev.to_dev(kr), dev.nullbits_dev(nl), dev.nullbits_dev(nr)
    for _ in range(2):
        k, c, f, j = dev.join_group_count(dl, dnl, dr, dnr)
        assert dev.last_join_form() == 0, "expected the 64-bit form"
        if expect_pruned:	# (the other way round nothing is promised: a verdict remembered for recycled buffer addresses may prune - exact all the same)
            assert dev.last_join_filter()[1], (case, dev.last_join_filter())
        assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec), case
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef), case


@pytest.mark.parametrize("shape", ["dup16_pruned", "unique_2e22", "nulls_offset", "group_only", "window_2e17_join", "window_2e18_group", "window_2e16_group"])
def test_key_windows_up_to_2e23_are_partitioned_once_and_joined_by_wide_direct_leaves(dev, narrow_mode, monkeypatch, shape):
    """Compact narrow form with a window of 2^16 ... 2^23 key values: ONE 9-bit partition level and k_leaf_wide (tables of
    2^(k - 9) entries, 16-bit row counts) instead of two levels and k_leaf_direct.  Same groups, counts, first rows and order
    as the oracle and as the two-level form (MDB_ONE_LEVEL=0)."""
    narrow_mode(1)
    monkeypatch.setenv("MDB_GROUP_BANDED", "0")     # (single-table GROUP BY from 2^21 rows on goes through the band sort by default: its own test)
    rng = np.random.default_rng(len(shape) * 101 + 7)
    nl = nr = None
    has_r = True
    if shape == "dup16_pruned":          # the benchmark's variant D in small: the right table holds the lowest sixteenth
        n_l = n_r = 6_000_000 + 8192
        kl = rng.permutation(16 * 5_000_000)[:n_l].astype(np.int64)
        kr = rng.integers(0, 5_000_000, n_r, dtype=np.int64)
    elif shape == "unique_2e22":
        n_l, n_r = 4_000_000, 3_500_000
        kl = rng.permutation(4_100_000)[:n_l].astype(np.int64)
        kr = rng.permutation(4_100_000)[:n_r].astype(np.int64)
    elif shape == "nulls_offset":
        n_l, n_r = 2_500_000, 1_800_000
        kl = rng.integers(0, 1_500_000, n_l, dtype=np.int64) - 2**40
        kr = rng.integers(0, 1_500_000, n_r, dtype=np.int64) - 2**40
        nl, nr = rng.random(n_l) < 0.05, rng.random(n_r) < 0.1
    elif shape == "window_2e17_join":      # tables of 2^8 entries per digit: a 10^5-key dimension, 30 fact rows per key
        n_l, n_r = 3_000_000, 100_000
        kl = rng.integers(0, 100_000, n_l, dtype=np.int64) + 7_000_000
        kr = rng.permutation(100_000).astype(np.int64) + 7_000_000
        nl = rng.random(n_l) < 0.02
    elif shape == "window_2e18_group":     # 10^5 distinct values, 40 rows each: duplicates in the key sample, one level all the same
        has_r = False
        n_l, n_r = 4_000_000, 0
        kl = rng.integers(0, 100_000, n_l, dtype=np.int64) * 2 - 100_000
        kr = None
    elif shape == "window_2e16_group":     # 6 * 10^4 distinct values, 50 rows each: tables of 128 entries
        has_r = False
        n_l, n_r = 3_000_000, 0
        kl = rng.integers(0, 60_000, n_l, dtype=np.int64) + 2**33
        kr = None
    else:
        has_r = False
        n_l, n_r = 7_000_000, 0
        kl = rng.integers(0, 3_000_000, n_l, dtype=np.int64) * 2 + 10**12
        kr = None
    dl, dnl = dev.to_dev(kl), dev.nullbits_dev(nl)
    if has_r:
        ek, ec, ef, ej = orc.join_group_count(kl, nl, kr, nr)
        dr, dnr = dev.to_dev(kr), dev.nullbits_dev(nr)
    else:
        ef, ec = orc.group_count(kl, nl)
        ek = ej = None
    for one_level in ("1", "0", "1"):
        monkeypatch.setenv("MDB_ONE_LEVEL", one_level)
        if has_r:
            k, c, f, j = dev.join_group_count(dl, dnl, dr, dnr)
            assert dev.last_join_form() == 2, shape
            assert j == ej, (shape, one_level)
            assert np.array_equal(_np(k), ek), (shape, one_level)
        else:
            f, c = dev.group_count(dl, dnl)
        assert dev.last_join_levels() == (1 if one_level == "1" else 2), (shape, one_level)
        assert np.array_equal(_np(c), ec), (shape, one_level)
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef), (shape, one_level)


@pytest.mark.parametrize("rows_of_the_key", [65_535, 65_536, 70_000])
def test_a_row_count_beyond_16_bits_sends_the_wide_direct_leaves_back_to_two_levels(dev, narrow_mode, rows_of_the_key, monkeypatch):
    """k_leaf_wide counts rows per key in 16-bit halves of LDS words.  A key with 2^16 or more rows carries into (or out
    of) its neighbour; the kernel notices that the sum of the counts falls short of the rows it counted, and the operator is
    redone with two partition levels.  1.6 * 10^8 rows (so that the one key's rows do not overflow a first-level region
    before the leaf kernel sees them), plain GROUP BY, checked on the device against torch: COUNT per key, first row per key,
    first-occurrence order.  65 535 rows still fit."""
    narrow_mode(1)
    monkeypatch.setenv("MDB_GROUP_BANDED", "0")     # (the partitioned path is what is tested here)
    n, span = 160_000_000, 4_000_000
    g = torch.Generator(device="cuda")
    g.manual_seed(rows_of_the_key)
    kl = torch.randint(0, span, (n,), dtype=torch.int64, device="cuda", generator=g)
    kl[kl == 777] = 778
    kl[torch.randperm(n, device="cuda", generator=g)[:rows_of_the_key]] = 777
    f, c = dev.group_count(kl, None)
    assert dev.last_join_levels() == (1 if rows_of_the_key < 65_536 else 2)
    fi = f.to(torch.int64) & 0xFFFFFFFF
    assert bool((fi[1:] > fi[:-1]).all())
    keys = kl[fi]
    uk, uc = torch.unique(kl, return_counts=True)
    o = torch.argsort(keys)
    assert torch.equal(keys[o], uk) and torch.equal(c[o], uc)
    assert int(c[keys == 777]) == rows_of_the_key
    first = torch.full((span,), n, dtype=torch.int64, device="cuda")
    first.scatter_reduce_(0, kl, torch.arange(n, device="cuda"), "amin")
    assert torch.equal(first[keys], fi)
    del kl, first, uk, uc, keys, o
    torch.cuda.empty_cache()


@pytest.mark.parametrize("rows_of_the_key", [65_535, 65_536, 70_000])
@pytest.mark.parametrize("side", ["right", "left"])
def test_a_row_count_beyond_16_bits_in_the_any_order_leaf_is_noticed_not_wrapped(dev, rows_of_the_key, side):
    """the any-order join + GROUP BY (no MDB_ORDER_FIRST; the sharded operator's receiver runs the same leaf) counts rows per key in 16-bit
    halves of LDS words with atomics that do not come back (round 6): a key with 2^16 rows or more carries into its neighbour or out of the
    word - noticed as fields that sum to less than the rows added (flag 2048), the operator then answers by another path.  Either way:
    every key's COUNT = its left rows x its right rows, checked on the device against torch; 65 535 rows still fit."""
    n, span = 40_000_000, 1 << 26
    g = torch.Generator(device="cuda")
    g.manual_seed(rows_of_the_key * (1 if side == "right" else 3))
    kr = torch.randint(0, span, (n,), dtype=torch.int64, device="cuda", generator=g) + 1_000_000_000
    kl = kr[torch.randint(0, n, (n,), device="cuda", generator=g)]
    hot, other = kr[12345].clone(), kr[12346].clone()
    tgt = kr if side == "right" else kl
    kr[kr == hot] = other
    kl[kl == hot] = other
    tgt[torch.randperm(n, device="cuda", generator=g)[:rows_of_the_key]] = hot
    if side == "right":
        kl[:5] = hot
    else:
        kr[:3] = hot
    dev.prof_enable(True)
    dev.prof_reset()
    k, c, j = dev.join_group_count_unordered(kl, None, kr, None)
    ran = set(dev.prof_read())
    dev.prof_enable(False)
    assert "shard_leaf_wide" in ran, ran        # (the leaf this test is about did run - and, beyond 65 535 rows, was not the one that answered)
    ul, cl = torch.unique(kl, return_counts=True)
    ur, cr = torch.unique(kr, return_counts=True)
    pos = torch.searchsorted(ur, ul).clamp(max=ur.numel() - 1)
    hit = ur[pos] == ul
    ek, ec = ul[hit], cl[hit] * cr[pos[hit]]
    o = torch.argsort(k)
    assert torch.equal(k[o], ek) and torch.equal(c[o], ec) and j == int(ec.sum())
    assert int(c[k == hot]) == rows_of_the_key * (5 if side == "right" else 3)
    del kl, kr, ul, ur, cl, cr, k, c
    torch.cuda.empty_cache()


@pytest.mark.parametrize("shape", ["spread", "bunched"])
def test_few_groups_are_ordered_through_row_id_bitmaps_or_the_general_sort_alike(dev, narrow_mode, monkeypatch, shape):
    """Few group records among many left rows (a selective join) are ordered by k_order_leaf_sparse - leaves of 2^16 row
    ids, ranks from a bitmap of the leaf's row ids - when a leaf's share fits a workgroup's registers.  'spread': the groups'
    first rows lie all over the table (the attempt succeeds); 'bunched': every group's first row is among the first 250 000
    rows, four leaves get 50 000 records each and the general ordering sort takes over.  Both equal the oracle and the
    result with the attempt switched off."""
    narrow_mode(1)
    rng = np.random.default_rng(2024 if shape == "spread" else 2025)
    n_l, groups = 16_000_000 + (4096 if shape == "spread" else 0), 200_000
    if shape == "spread":
        kl = rng.permutation(n_l).astype(np.int64)
        kr = rng.choice(n_l, groups, replace=False).astype(np.int64)
    else:
        kl = rng.integers(0, groups, n_l, dtype=np.int64)
        kl[:groups] = rng.permutation(groups)
        kr = np.arange(groups, dtype=np.int64)
    ek, ec, ef, ej = orc.join_group_count(kl, None, kr, None)
    assert ek.size == groups
    if shape == "bunched":
        assert int(ef.max()) < 250_000
    dl, dr = dev.to_dev(kl), dev.to_dev(kr)
    for sparse in ("1", "0", "1"):
        monkeypatch.setenv("MDB_ORDER_SPARSE", sparse)
        k, c, f, j = dev.join_group_count(dl, None, dr, None)
        assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec), (shape, sparse)
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef), (shape, sparse)


def test_min_max_pruning_with_a_right_table_of_nothing_but_null_keys(dev, narrow_mode, monkeypatch):
    """No right key at all: an empty join, as the reference's - a NULL key joins nothing (executor_select.c:557-579); with
    the right table forced first (MDB_MINMAX_PRUNE=2: prune whatever the key sample says) the recorded range stays empty
    (smallest > largest) and every left row is dropped.  One real right key at the far end of the left table's range brings
    back exactly its group."""
    narrow_mode(1)
    monkeypatch.setenv("MDB_MINMAX_PRUNE", "2")
    rng = np.random.default_rng(77)
    n_l, n_r = 2_222_222, 700_000
    kl = rng.permutation(n_l).astype(np.int64)
    kr = rng.integers(0, n_l // 8, n_r, dtype=np.int64)
    nr = np.ones(n_r, dtype=bool)
    dl, dr = dev.to_dev(kl), dev.to_dev(kr)
    k, c, f, j = dev.join_group_count(dl, None, dr, dev.nullbits_dev(nr))
    assert k.numel() == 0 and j == 0
    nr[123] = False
    kr[123] = n_l - 1
    ek, ec, ef, ej = orc.join_group_count(kl, None, kr, nr)
    k, c, f, j = dev.join_group_count(dl, None, dev.to_dev(kr), dev.nullbits_dev(nr))
    assert ej == 1 and j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec)
    assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef)


@pytest.mark.parametrize("shape", ["dup16", "subrange", "few_right_rows", "fifth_of_the_rows_nulls", "no_match", "nulls", "offset"])
def test_left_table_pruning_by_the_right_tables_keys_does_not_change_results(dev, narrow_mode, monkeypatch, shape):
    """Compact narrow form, unsplit call: the right table is partitioned first.  Min-max pruning - its first level records
    the exact key range, the left table's first level drops the rows outside - and, for right tables that are small but
    spread over the whole range, the semi-join bitmap at the second level (exact, and one bit per 2 / 4 / 8 adjacent hashed
    values).  Groups, counts, first rows, order and joined rows equal the oracle's with every combination on and off.  (The
    bitmap belongs to the two-level form: these 2^22-value windows take ONE partition level unless MDB_ONE_LEVEL=0 - the
    last combination - where the leaf kernel reads the left rows only once anyway.)"""
    narrow_mode(1)
    rng = np.random.default_rng(len(shape) * 13 + 5)
    # (every shape its own table sizes: the operator remembers what it learned about a column by address and length, and
    #  the allocator hands the next case the same addresses)
    n_l, n_r, span, off = 3_000_000 + 4096 * len(shape), 2_500_000 + 4096 * len(shape), 3_000_000 + 4096 * len(shape), 0
    kl = rng.permutation(span)[:n_l].astype(np.int64)
    nl = nr = None
    by_rows = shape in ("few_right_rows", "fifth_of_the_rows_nulls")
    if shape == "dup16":
        kr = (rng.permutation(span)[:n_r] % (span // 16)).astype(np.int64)
    elif shape == "subrange":
        kr = rng.integers(span // 3, span // 3 + span // 6, n_r, dtype=np.int64)
    elif shape == "few_right_rows":
        kr = rng.integers(0, span, 40_000, dtype=np.int64)
    elif shape == "fifth_of_the_rows_nulls":
        kr = rng.integers(0, span, n_l // 5, dtype=np.int64)
        nl, nr = rng.random(n_l) < 0.05, rng.random(kr.size) < 0.3
    elif shape == "no_match":
        kl = 2 * rng.integers(0, span // 2, n_l, dtype=np.int64)
        kr = 2 * rng.integers(0, span // 12, n_r, dtype=np.int64) + 1
    elif shape == "nulls":
        kr = rng.integers(0, span // 10, n_r, dtype=np.int64)
        nl, nr = rng.random(n_l) < 0.05, rng.random(kr.size) < 0.3
    else:
        off = -(2**45)
        kr = rng.integers(0, span // 5, n_r, dtype=np.int64)
    kl, kr = kl + off, kr + off
    ek, ec, ef, ej = orc.join_group_count(kl, nl, kr, nr)
    dl, dr, dnl, dnr = dev.to_dev(kl), dev.to_dev(kr), dev.nullbits_dev(nl), dev.nullbits_dev(nr)
    for prune, on, slice_bits, expect, one in ((None, None, None, 1, "0"), (None, "1", "13", 2, "0"), (None, "1", "12", 3, "0"), (None, "1", "11", 4, "0"),
                                               (None, "1", "9", 0, "0"), (None, "0", None, 0, "0"), ("0", None, None, 0, "0"), (None, None, None, 0, None),
                                               ("0", None, None, 0, None)):
        for name, val in (("MDB_MINMAX_PRUNE", prune), ("MDB_SEMIJOIN", on), ("MDB_SEMIJOIN_SLICE", slice_bits), ("MDB_ONE_LEVEL", one)):
            if val is None:
                monkeypatch.delenv(name, raising=False)
            else:
                monkeypatch.setenv(name, val)
        k, c, f, j = dev.join_group_count(dl, dnl, dr, dnr)
        assert dev.last_join_form() == 2, (shape, prune, on, slice_bits)
        assert dev.last_join_levels() == (2 if one == "0" else 1), (shape, prune, on, slice_bits)
        assert dev.last_join_filter() == ((expect if by_rows else 0) if prune is None else 0, prune is None), (shape, prune, on, slice_bits)
        assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec), (shape, prune, on, slice_bits)
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef), (shape, prune, on, slice_bits)


@pytest.mark.parametrize("shape", ["low_sixteenth", "middle", "negative_window", "nulls", "nothing_in_range"])
def test_min_max_pruning_in_the_plain_narrow_form(dev, narrow_mode, monkeypatch, shape):
    """Keys too sparse for the compact form (3 * 10^6 rows over a 2^31-wide range: hashed leaf tables, 32-bit hashes) are
    pruned the same way: the right table's exact key range, recorded by its first partition level, drops the left rows
    outside at theirs.  Same groups, counts, first rows and order as the oracle, with the pruning on and off."""
    narrow_mode(1)
    rng = np.random.default_rng(len(shape) * 19 + 1)
    n_l, n_r = 3_000_000 + 4096 * len(shape), 2_000_000 + 4096 * len(shape)
    lo, hi = 0, 2**31 - 5
    kl = rng.integers(lo, hi, n_l, dtype=np.int64)
    nl = nr = None
    if shape == "low_sixteenth":
        pool = rng.choice(kl[kl < hi // 16], 300_000)
    elif shape == "middle":
        pool = rng.choice(kl[(kl > hi // 3) & (kl < hi // 3 + hi // 10)], 300_000)
    elif shape == "negative_window":
        kl = kl - 2**40 - 2**30
        pool = rng.choice(kl[kl < -(2**40) - 2**30 + hi // 20], 300_000)
    elif shape == "nulls":
        pool = rng.choice(kl[kl < hi // 8], 300_000)
        nl, nr = rng.random(n_l) < 0.05, rng.random(n_r) < 0.2
    else:
        kl = kl >> 1
        pool = 2**30 + 7 + rng.integers(0, 2**24, 300_000)	# right keys beyond every left key (still in the 2^31-wide window)
    kr = pool[rng.integers(0, pool.size, n_r)]
    ek, ec, ef, ej = orc.join_group_count(kl, nl, kr, nr)
    dl, dr, dnl, dnr = dev.to_dev(kl), dev.to_dev(kr), dev.nullbits_dev(nl), dev.nullbits_dev(nr)
    for prune in (None, "0", None):
        if prune is None:
            monkeypatch.delenv("MDB_MINMAX_PRUNE", raising=False)
        else:
            monkeypatch.setenv("MDB_MINMAX_PRUNE", prune)
        k, c, f, j = dev.join_group_count(dl, dnl, dr, dnr)
        assert dev.last_join_form() == 1 and dev.last_join_filter() == (0, prune is None), (shape, prune, dev.last_join_form(), dev.last_join_filter())
        assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec), (shape, prune)
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef), (shape, prune)


@pytest.mark.parametrize("shape", ["dup16", "subrange", "nulls", "offset"])
def test_keyed_group_records_decode_the_key_instead_of_gathering_it(dev, narrow_mode, monkeypatch, shape):
    """Selective joins in the compact narrow form write (first row, hashed key, COUNT) records and the ordering kernel
    decodes the group key from them (mdb_unmixk) instead of gathering it from the key column: identical keys, counts,
    first rows and order with the records keyed and plain, against the oracle."""
    narrow_mode(1)
    rng = np.random.default_rng(len(shape) * 17 + 3)
    n_l, n_r, span, off = 3_100_000 + 8192 * len(shape), 2_400_000 + 8192 * len(shape), 3_100_000 + 8192 * len(shape), 0
    kl = rng.permutation(span)[:n_l].astype(np.int64)
    nl = nr = None
    if shape == "dup16":
        kr = (rng.permutation(span)[:n_r] % (span // 16)).astype(np.int64)
    elif shape == "subrange":
        kr = rng.integers(span // 3, span // 3 + span // 6, n_r, dtype=np.int64)
    elif shape == "nulls":
        kr = rng.integers(0, span // 10, n_r, dtype=np.int64)
        nl, nr = rng.random(n_l) < 0.05, rng.random(kr.size) < 0.3
    elif shape == "offset":
        off = 10**15
        kr = rng.integers(0, span // 5, n_r, dtype=np.int64)
    else:
        off = 10**15
        kr = rng.integers(0, span // 5, n_r, dtype=np.int64)
    kl, kr = kl + off, kr + off
    ek, ec, ef, ej = orc.join_group_count(kl, nl, kr, nr)
    dl, dr, dnl, dnr = dev.to_dev(kl), dev.to_dev(kr), dev.nullbits_dev(nl), dev.nullbits_dev(nr)
    for keyed in ("1", "0", "1"):
        monkeypatch.setenv("MDB_KEYED_RECORDS", keyed)
        k, c, f, j = dev.join_group_count(dl, dnl, dr, dnr)
        assert dev.last_join_form() == 2, (shape, keyed)
        assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec), (shape, keyed)
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef), (shape, keyed)


def test_four_byte_group_records_on_a_remembered_verdict_and_its_retraction(dev, narrow_mode):
    """The second join over the same columns writes 4-byte group records straight from the leaf kernel (the first run saw every
    COUNT(*) fit beside the row id).  When the buffers then change under the same addresses so that a COUNT no longer fits
    (one key: 4 left x 300 right rows = 1200 >= 2^10 at 22 row-id bits), the kernel reports it and the operator is redone with
    8-byte records: always the oracle's result."""
    narrow_mode(1)
    rng = np.random.default_rng(5150)
    n = 3_000_000
    kl = rng.permutation(n).astype(np.int64)
    kr = rng.permutation(n).astype(np.int64)		# same range, unique keys: no pruning, no keyed records
    dl, dr = dev.to_dev(kl), dev.to_dev(kr)
    ek, ec, ef, ej = orc.join_group_count(kl, None, kr, None)
    for _ in range(3):
        k, c, f, j = dev.join_group_count(dl, None, dr, None)
        assert dev.last_join_form() == 2 and j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec)
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef)
    kl[rng.choice(n, 4, replace=False)] = 777
    kr[rng.choice(n, 300, replace=False)] = 777
    dl.copy_(torch.from_numpy(kl).to(dl.device))
    dr.copy_(torch.from_numpy(kr).to(dr.device))
    ek, ec, ef, ej = orc.join_group_count(kl, None, kr, None)
    for _ in range(2):
        k, c, f, j = dev.join_group_count(dl, None, dr, None)
        assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec)
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef)
    assert int(ec.max()) >= 1200


def test_keyed_group_records_give_way_when_a_count_does_not_fit(dev, narrow_mode, monkeypatch):
    """4 * 10^7 x 4 * 10^7 rows (26 row-id bits + 26 key bits leave 12 bits of a keyed record for COUNT(*)): one key that
    8 left rows and 512 right rows hold has COUNT(*) = 4096 - the kernel reports it, the operator is redone with plain
    records and remembers; results equal the plain-record run element for element."""
    narrow_mode(1)
    n = 40_000_000
    kl = dev.gen_keys(n, 0, n, 42, 0)
    kr = dev.gen_keys(n, 0, n, 43, n // 16)
    key = int(kl[12345].item()) % (n // 16)		# a value both tables hold
    pos_l = torch.arange(8, device=kl.device) * 4_000_001 + 77
    pos_r = torch.arange(512, device=kl.device) * 70_001 + 5
    kl[pos_l] = key
    kr[pos_r] = key
    out = {}
    for keyed in ("0", "1", "1"):
        monkeypatch.setenv("MDB_KEYED_RECORDS", keyed)
        k, c, f, j = dev.join_group_count(kl, None, kr, None)
        assert dev.last_join_form() == 2
        out[keyed] = (k.clone(), c.clone(), f.clone(), j)
    for a, b in zip(out["0"][:3], out["1"][:3]):
        assert torch.equal(a, b)
    assert out["0"][3] == out["1"][3]
    k, c = out["1"][0], out["1"][1]
    assert int(c[k == key].item()) >= 4096


def test_min_max_pruning_at_scale_matches_the_unpruned_operator(dev, narrow_mode, monkeypatch):
    """4 * 10^7 x 4 * 10^7 rows of the benchmark's variant D (1 left row in 16 has a partner): identical columns with the
    filter on and off; variant U (every row has one) does not take the filter."""
    narrow_mode(1)
    n = 40_000_000
    kl = dev.gen_keys(n, 0, n, 42, 0)
    kr = dev.gen_keys(n, 0, n, 43, n // 16)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MDB_MINMAX_PRUNE", mode)
        k, c, f, j = dev.join_group_count(kl, None, kr, None)
        assert dev.last_join_filter() == (0, mode == "1")		# the right table's SPAN is small: pruned by range, no bitmap
        out[mode] = (k.clone(), c.clone(), f.clone(), j)
    for a, b in zip(out["0"][:3], out["1"][:3]):
        assert torch.equal(a, b)
    assert out["0"][3] == out["1"][3] == n
    monkeypatch.delenv("MDB_MINMAX_PRUNE")
    kr = dev.gen_keys(n, 0, n, 43, 0)
    k, c, f, j = dev.join_group_count(kl, None, kr, None)
    assert dev.last_join_form() == 2 and dev.last_join_filter() == (0, False) and j == n and k.numel() == n	# same key range: nothing to prune


@pytest.mark.parametrize("packed", ["1", "0"])
@pytest.mark.parametrize("n", [262_144, 600_001])
def test_group_count_multi_and_distinct_on_the_packed_sort_path_unpinned(dev, n, packed, monkeypatch):
    """From 2^18 rows on, INT64 columns whose ranges fit one word are sorted by the packed path and the group / distinct
    run heads come from its sorted composite values (no per-row column gathers): against the numpy oracle with NULLs,
    negative values, a row-id vector, many and few groups; a DOUBLE column keeps the general path."""
    monkeypatch.setenv("MDB_GROUP_MULTI_PACKED", packed)
    rng = np.random.default_rng(n + 31)
    a = rng.integers(-50, 50, n, dtype=np.int64)
    b = rng.integers(10**12, 10**12 + 300, n, dtype=np.int64)
    c = rng.integers(0, 2, n, dtype=np.int64)
    x = np.round(rng.normal(0, 1, n), 0)
    na, nb = rng.random(n) < 0.1, rng.random(n) < 0.3
    rid = rng.integers(0, n, n).astype(np.uint32)
    ad, bd, cd, xd = dev.to_dev(a), dev.to_dev(b), dev.to_dev(c), dev.to_dev(x)
    nad, nbd, ridd = dev.nullbits_dev(na), dev.nullbits_dev(nb), dev.to_dev(rid)
    for keys_np, keys_dev in [
            ([(a, na, None, False, False), (b, nb, None, False, False)], [(ad, nad, None, D.T_INT64, False), (bd, nbd, None, D.T_INT64, False)]),
            ([(c, None, None, False, False)], [(cd, None, None, D.T_INT64, False)]),
            ([(b, None, rid, False, False), (a, na, rid, False, False), (c, None, rid, False, False)],
             [(bd, None, ridd, D.T_INT64, False), (ad, nad, ridd, D.T_INT64, False), (cd, None, ridd, D.T_INT64, False)]),
            ([(a, None, None, False, False), (x, None, None, True, False)], [(ad, None, None, D.T_INT64, False), (xd, None, None, D.T_DOUBLE, False)])]:
        first, cnt = dev.group_count_multi(keys_dev, n)
        ef, ec = orc.group_count_multi(keys_np, n)
        assert np.array_equal(_np(first).view(np.uint32).astype(np.int64), ef) and np.array_equal(_np(cnt), ec)
        got = _np(dev.distinct_sel(keys_dev, n)).view(np.uint32)
        assert np.array_equal(got, orc.distinct_sel(keys_np, n))


@pytest.mark.parametrize("shape", ["thousand", "two", "one", "negative", "span_4096", "span_4097", "span_10000_nulls", "span_13000", "span_13500",
                                   "span_12000_unsampled_extremes", "outlier", "all_null", "half_null", "huge_offset", "sorted_runs"])
@pytest.mark.parametrize("n", [262_144, 1_300_001])
def test_group_count_over_a_small_value_range(dev, shape, n):
    """Plain GROUP BY + COUNT(*) whose key column spans few values (the direct LDS-table path from 2^18 rows on - up to 4096
    values in a replicated 64 KiB table, up to 13 000 in a 128 KiB one with a window of 1.25 x the sampled span; a value
    outside the window or a wider span falls back to the partitioned path): groups in first-occurrence order, NULL group
    included, against the numpy oracle."""
    rng = np.random.default_rng(n + len(shape))
    nulls = None
    if shape == "thousand":
        k = rng.integers(0, 1000, n)
    elif shape == "two":
        k = rng.integers(0, 2, n)
    elif shape == "one":
        k = np.full(n, 7)
    elif shape == "negative":
        k = rng.integers(-300, -100, n)
    elif shape == "span_4096":
        k = rng.integers(5000, 5000 + 4096, n)
        k[:2] = [5000, 5000 + 4095]
    elif shape == "span_4097":
        k = rng.integers(5000, 5000 + 4097, n)
        k[:2] = [5000, 5000 + 4096]
    elif shape == "span_10000_nulls":
        k = rng.integers(-4000, 6000, n)
        nulls = rng.random(n) < 0.1
    elif shape == "span_13000":
        k = rng.integers(10**12, 10**12 + 13000, n)
    elif shape == "span_13500":
        k = rng.integers(0, 13500, n)
    elif shape == "span_12000_unsampled_extremes":    # two values beyond the window of 1.25 x the sampled span
        k = rng.integers(0, 12000, n)
        k[n // 2 + 1] = -4000
        k[n // 3 + 1] = 17000
    elif shape == "outlier":
        k = rng.integers(0, 50, n)
        k[n // 2 + 1] = 10**9                       # not among the sampled rows
        k[n // 3 + 1] = -70
    elif shape == "all_null":
        k = rng.integers(0, 10, n)
        nulls = np.ones(n, dtype=bool)
    elif shape == "half_null":
        k = rng.integers(0, 10, n)
        nulls = rng.random(n) < 0.5
        nulls[:3] = [False, True, False]
    elif shape == "huge_offset":
        k = 2**62 + rng.integers(0, 600, n)
    else:
        k = np.repeat(np.arange(37)[::-1], n // 37 + 1)[:n]
    k = np.asarray(k, dtype=np.int64)
    first, cnt = dev.group_count(dev.to_dev(k), dev.nullbits_dev(nulls))
    e_first, e_cnt = orc.group_count(k, nulls)
    assert np.array_equal(_np(first).view(np.uint32).astype(np.int64), e_first) and np.array_equal(_np(cnt), e_cnt)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("n_l,n_r", [(5, 40), (3000, 50_000), (250_000, 900_000), (1_200_000, 2_500_000)])
def test_join_pairs_unique_left_keys_duplicate_right_keys(dev, narrow_mode, mode, n_l, n_r):
    """FROM pk_table JOIN fk_table: unique keys on the LEFT, many rows per key on the right - the unique-key join runs
    with the sides swapped and a stable sort restores the reference's left-major / right-minor pair order.  Repeated on
    the same columns (the remembered verdicts are used), with NULLs on both sides and unmatched keys."""
    narrow_mode(mode)
    rng = np.random.default_rng(n_l * 7 + n_r + mode)
    kl = rng.permutation(2 * n_l)[:n_l].astype(np.int64) - n_l // 2                  # unique, half of them unmatched
    kr = rng.integers(-n_l // 2, n_l, n_r, dtype=np.int64)                           # ~n_r / 1.5 n_l rows per key
    nl = rng.random(n_l) < 0.02
    nr = rng.random(n_r) < 0.02
    el, er = orc.join_pairs(kl, nl, kr, nr)
    dl, dnl, dr, dnr = dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr), dev.nullbits_dev(nr)
    for _ in range(3):
        l, r = dev.join_pairs(dl, dnl, dr, dnr)
        assert l.numel() == len(el)
        assert np.array_equal(_np(l).astype(np.int64), el) and np.array_equal(_np(r).astype(np.int64), er)
    # the same buffers, now with a duplicate on the left as well: a true N:M join through the general path
    kl2 = kl.copy()
    kl2[n_l // 2] = kl2[0]
    nl2 = nl.copy()
    nl2[[0, n_l // 2]] = False
    dl.copy_(torch.from_numpy(kl2))
    dnl2 = dev.nullbits_dev(nl2)
    el, er = orc.join_pairs(kl2, nl2, kr, nr)
    l, r = dev.join_pairs(dl, dnl2, dr, dnr)
    assert np.array_equal(_np(l).astype(np.int64), el) and np.array_equal(_np(r).astype(np.int64), er)


@pytest.mark.parametrize("shape", ["perm_both", "subset_nulls", "offset_2e24", "fk_to_pk", "one_duplicate_right_key", "unsampled_key_outside"])
def test_join_pairs_unique_keys_in_a_window_up_to_2e24_take_one_partition_level(dev, narrow_mode, monkeypatch, shape):
    """Unique right keys inside a compact window of at most 2^24 values (BASELINE configs[1]'s primary-key join): one 9-bit
    partition level per table, a direct-address LDS table of right row ids per digit, the partner written to match[left row],
    the pairs compacted out of match[] in left-row order.  Same pairs, same order as the oracle and as the two-level path
    (MDB_ONE_LEVEL=0); a duplicated right key (noticed: fewer occupied entries than right rows) and a key outside the
    sampled window (noticed by the first partition level) go through the other paths."""
    narrow_mode(1)
    rng = np.random.default_rng(len(shape) * 31 + 3)
    nl = nr = None
    if shape == "perm_both":
        n_l, n_r = 3_000_000, 3_000_000
        kl = rng.permutation(n_l).astype(np.int64)
        kr = rng.permutation(n_r).astype(np.int64)
    elif shape == "subset_nulls":
        n_l, n_r = 2_200_000, 1_300_000
        kl = rng.integers(0, 4_000_000, n_l, dtype=np.int64)
        kr = rng.permutation(4_000_000)[:n_r].astype(np.int64)
        nl, nr = rng.random(n_l) < 0.03, rng.random(n_r) < 0.05
    elif shape == "offset_2e24":
        n_l, n_r = 5_000_000, 9_000_000
        kl = rng.integers(0, 16_000_000, n_l, dtype=np.int64) - 2**44
        kr = rng.permutation(16_000_000)[:n_r].astype(np.int64) - 2**44
    elif shape == "fk_to_pk":           # unique on the LEFT: the sides are swapped, then a stable sort
        n_l, n_r = 1_500_000, 4_000_000
        kl = rng.permutation(2_000_000)[:n_l].astype(np.int64)
        kr = rng.integers(0, 2_000_000, n_r, dtype=np.int64)
    elif shape == "one_duplicate_right_key":
        n_l, n_r = 2_000_000, 2_000_001
        kl = rng.integers(0, 2_000_000, n_l, dtype=np.int64)
        kr = np.concatenate([rng.permutation(2_000_000), [777]]).astype(np.int64)
    else:
        n_l, n_r = 2_500_000, 2_500_000
        kl = rng.permutation(n_l).astype(np.int64)
        kr = rng.permutation(n_r).astype(np.int64)
        kl[1_234_567] = 2**40          # (one row in 2.5 * 10^6: the 4096-key sample does not see it)
    el, er = orc.join_pairs(kl, nl, kr, nr)
    dl, dnl, dr, dnr = dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr), dev.nullbits_dev(nr)
    for one_level in ("1", "0", "1"):
        monkeypatch.setenv("MDB_ONE_LEVEL", one_level)
        l, r = dev.join_pairs(dl, dnl, dr, dnr)
        assert l.numel() == len(el), (shape, one_level)
        assert np.array_equal(_np(l).astype(np.int64), el) and np.array_equal(_np(r).astype(np.int64), er), (shape, one_level)


@pytest.mark.parametrize("shape", ["hundred", "one_key", "disjoint", "outlier_left", "outlier_right", "nulls", "offset", "span_too_wide", "span_11000"])
def test_join_group_count_over_a_small_value_range(dev, shape):
    """Join + GROUP BY join key + COUNT(*) whose key columns both lie in one window of at most 4096 values (joins on a few
    hot values: N:M counts in the billions) take the direct LDS-table path; an unsampled value outside the window or a
    wider span sends the operator through the partitioned path.  Keys, counts, first rows, joined-row total, order."""
    rng = np.random.default_rng(len(shape) * 3)
    n_l, n_r = 700_000, 1_100_000
    nl = nr = None
    if shape == "hundred":
        kl, kr = rng.integers(0, 100, n_l), rng.integers(20, 140, n_r)
    elif shape == "one_key":
        kl, kr = np.full(n_l, 5), np.full(n_r, 5)
    elif shape == "disjoint":
        kl, kr = rng.integers(0, 50, n_l), rng.integers(60, 90, n_r)
    elif shape == "outlier_left":
        kl, kr = rng.integers(0, 100, n_l), rng.integers(0, 100, n_r)
        kl[n_l // 2 + 1] = 10**7
        kr[9] = 10**7
    elif shape == "outlier_right":
        kl, kr = rng.integers(0, 100, n_l), rng.integers(0, 100, n_r)
        kr[n_r // 2 + 1] = -5000
        kl[3] = -5000
    elif shape == "nulls":
        kl, kr = rng.integers(0, 30, n_l), rng.integers(0, 30, n_r)
        nl, nr = rng.random(n_l) < 0.3, rng.random(n_r) < 0.6
    elif shape == "offset":
        kl, kr = -(2**50) + rng.integers(0, 3000, n_l), -(2**50) + rng.integers(0, 3000, n_r)
    elif shape == "span_11000":
        kl, kr = rng.integers(0, 11000, n_l) - 3000, rng.integers(2000, 9000, n_r) - 3000
        nl = rng.random(n_l) < 0.05
    else:
        kl, kr = rng.integers(0, 5000, n_l), rng.integers(0, 5000, n_r)
    kl, kr = np.asarray(kl, dtype=np.int64), np.asarray(kr, dtype=np.int64)
    ek, ec, ef, ej = orc.join_group_count(kl, nl, kr, nr)
    for _ in range(2):
        k, c, f, j = dev.join_group_count(dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr), dev.nullbits_dev(nr))
        assert j == ej
        assert np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec)
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef)


@pytest.mark.parametrize("d", [1, 7, 1000, 1150, 2600, 9000])
@pytest.mark.parametrize("n", [262_144, 1_500_000])
def test_group_count_few_distinct_values_anywhere_in_the_int64_range(dev, d, n):
    """Plain GROUP BY + COUNT(*) over few distinct values that are NOT close together (the hashed per-workgroup tables; the
    sample decides, a workgroup table that fills up sends the operator to the partitioned path): first-occurrence order,
    NULL group, the value 0 (whose hash is 0), against the numpy oracle.  Twice, so that remembered verdicts are used."""
    rng = np.random.default_rng(n + d)
    vals = np.unique(np.concatenate([[0, -1, np.iinfo(np.int64).min, np.iinfo(np.int64).max],
                                     rng.integers(np.iinfo(np.int64).min, np.iinfo(np.int64).max, d, dtype=np.int64)]))[:max(d, 1)]
    if d >= 2:
        vals[0] = 0
    k = vals[rng.integers(0, len(vals), n)]
    k[:3] = vals[-1]
    nulls = (rng.random(n) < 0.05) if d != 7 else None
    e_first, e_cnt = orc.group_count(k, nulls)
    kd, nd = dev.to_dev(k), dev.nullbits_dev(nulls)
    for _ in range(2):
        first, cnt = dev.group_count(kd, nd)
        assert np.array_equal(_np(first).view(np.uint32).astype(np.int64), e_first) and np.array_equal(_np(cnt), e_cnt)


@pytest.mark.parametrize("n_l,n_r", [(1, 2048), (1023, 1), (1024, 1025), (2047, 2048), (2048, 2048), (2048, 2049), (2049, 7)])
def test_single_workgroup_path_for_tiny_inputs_and_its_size_boundary(dev, n_l, n_r):
    """Up to 2048 rows per table the whole operator is one kernel of one workgroup (groups leave in first-occurrence order by
    a prefix sum over the left rows); one row more and the partitioned path runs.  Same results either side of the boundary:
    NULLs on both sides, the value 0 (hash 0), negative values, heavy duplicates, plain GROUP BY with its NULL group."""
    rng = np.random.default_rng(n_l * 5 + n_r)
    kl = rng.integers(-3, 40, n_l, dtype=np.int64)
    kr = rng.integers(-3, 60, n_r, dtype=np.int64)
    nl = rng.random(n_l) < 0.1
    nr = rng.random(n_r) < 0.1
    _jgc_check(dev, kl, nl, kr, nr)
    _jgc_check(dev, np.arange(n_l, dtype=np.int64)[::-1].copy(), None, rng.integers(0, max(n_l, 1), n_r, dtype=np.int64), None)


@pytest.mark.parametrize("n_l,n_r,dom", [(1, 1, 1), (3, 2, 3), (6, 6, 3), (100, 300, 40), (2048, 31, 5000), (2048, 2048, 100_000),
                                         (2048, 2048, 40), (2049, 10, 5), (300, 300, 1)])
def test_single_workgroup_materialising_join_and_its_boundaries(dev, n_l, n_r, dom):
    """Up to 2048 rows per table the materialising join is one single-workgroup kernel (the right keys in LDS, every left row
    walks over them: the reference's nested loop, its pair order by construction); more than 65536 pairs, or one row more,
    and the partitioned paths run.  NULLs, negative values, N:M duplicates."""
    rng = np.random.default_rng(n_l * 3 + n_r + dom)
    kl = rng.integers(-2, dom, n_l, dtype=np.int64)
    kr = rng.integers(-2, dom, n_r, dtype=np.int64)
    nl = rng.random(n_l) < 0.15
    nr = rng.random(n_r) < 0.15
    el, er = orc.join_pairs(kl, nl, kr, nr)
    l, r = dev.join_pairs(dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr), dev.nullbits_dev(nr))
    assert l.numel() == len(el)
    assert np.array_equal(_np(l).astype(np.int64), el) and np.array_equal(_np(r).astype(np.int64), er)


@pytest.mark.parametrize("specs", [
    [("full", False, 0.0)], [("full", True, 0.2)], [("double", False, 0.0)], [("double", True, 0.1)], [("const", True, 0.5)],
    [("small", False, 0.0), ("neg", True, 0.0)], [("small", True, 0.3), ("double", False, 0.3), ("small", False, 0.0)],
    [("small", False, 1.0)], [("neg", False, 0.0), ("full", False, 0.0), ("double", True, 0.0), ("small", True, 0.2)],
    [("small", False, 0.0), ("neg", True, 0.0), ("small", True, 0.0), ("neg", False, 0.0), ("small", False, 0.0)],
], ids=lambda s: "+".join(f"{k}{'D' if d else 'A'}{int(nf * 10)}" for k, d, nf in s))
@pytest.mark.parametrize("n", [2, 777, 2048, 2049])
def test_sort_perm_single_workgroup_path_for_tiny_inputs_unpinned(dev, specs, n):
    """Up to 2048 rows and four columns the permutation comes from one workgroup that ranks every row by counting; five
    columns, or one row more, take the radix passes.  Same stable order (NULLs first ascending / last descending, DOUBLE
    by total order) either way."""
    rng = np.random.default_rng(n + len(specs))
    _sort_case(dev, rng, n, specs)
    _sort_case(dev, rng, n, specs, with_rid=True)


@pytest.mark.parametrize("n", [1, 63, 129, 100_003, 1_000_001])
def test_filter_one_column_term_lists(dev, n):
    """Several comparisons of ONE INT64 column with constants, all ANDed (ranges) or all ORed (IN lists) - the specialised
    kernels - against the numpy predicate oracle: every comparison operator, constant on either side, NULLs (a NULL fails
    every term), up to 8 terms; 9 terms, a second column or mixed AND / OR go through the interpreter with the same result."""
    rng = np.random.default_rng(n)
    a = rng.integers(-50, 50, n, dtype=np.int64)
    b = rng.integers(-50, 50, n, dtype=np.int64)
    na = rng.random(n) < 0.1
    cols_np = [(a, na, None), (b, None, None)]
    cols_dev = [(dev.to_dev(a), dev.nullbits_dev(na), None), (dev.to_dev(b), None, None)]
    C, K = D.P_CMP_COL_CONST, D.P_CMP_CONST_COL
    AND, OR = (D.P_AND, 0, 0, 0, 0, 0), (D.P_OR, 0, 0, 0, 0, 0)
    progs = [
        [(C, D.CMP_GE, D.T_INT64, 0, 0, -10), (C, D.CMP_LE, D.T_INT64, 0, 0, 30), AND],
        [(K, D.CMP_LT, D.T_INT64, 0, 0, -10), (C, D.CMP_LT, D.T_INT64, 0, 0, 30), AND, (C, D.CMP_NE, D.T_INT64, 0, 0, 5), AND],
        [(C, D.CMP_GT, D.T_INT64, 0, 0, 0), (C, D.CMP_NE, D.T_INT64, 0, 0, 7), (K, D.CMP_GE, D.T_INT64, 0, 0, 40), AND, AND],
        [(C, D.CMP_EQ, D.T_INT64, 0, 0, v) for v in (1, 2, 3, -49, 49)] + [OR] * 4,
        [(C, D.CMP_EQ, D.T_INT64, 0, 0, 1), (C, D.CMP_EQ, D.T_INT64, 0, 0, 2), OR, (C, D.CMP_GT, D.T_INT64, 0, 0, 45), OR],
        [(C, D.CMP_EQ, D.T_INT64, 0, 0, v) for v in range(8)] + [OR] * 7,
        [(D.P_ISNULL, 1, 0, 0, 0, 0), (C, D.CMP_NE, D.T_INT64, 0, 0, 5), AND, (C, D.CMP_GE, D.T_INT64, 0, 0, 3), AND],      # IS NOT NULL AND ...
        [(C, D.CMP_EQ, D.T_INT64, 0, 0, 1), (C, D.CMP_EQ, D.T_INT64, 0, 0, 2), OR, (D.P_ISNULL, 0, 0, 0, 0, 0), OR],        # IN (...) OR IS NULL
        [(C, D.CMP_EQ, D.T_INT64, 0, 0, 1), (D.P_ISNULL, 0, 0, 0, 0, 0), AND],                                              # IS NULL in an AND list: interpreter
        [(C, D.CMP_EQ, D.T_INT64, 0, 0, 1), (D.P_ISNULL, 1, 0, 0, 0, 0), OR],                                              # IS NOT NULL in an OR list: interpreter
        [(C, D.CMP_EQ, D.T_INT64, 0, 0, v) for v in range(9)] + [OR] * 8,                                  # 9 terms: interpreter
        [(C, D.CMP_GE, D.T_INT64, 0, 0, -10), (C, D.CMP_LE, D.T_INT64, 1, 0, 30), AND],                   # two columns: interpreter
        [(C, D.CMP_EQ, D.T_INT64, 0, 0, 1), (C, D.CMP_EQ, D.T_INT64, 0, 0, 2), OR, (C, D.CMP_LT, D.T_INT64, 0, 0, 2), AND],  # mixed
    ]
    for prog in progs:
        exp = orc.filter_positions(prog, cols_np, n)
        got = _np(dev.filter(prog, cols_dev, n)).astype(np.int64)
        assert np.array_equal(got, exp), prog


def test_column_memo_keeps_several_table_pairs(dev):
    """What the operators learn about key columns is kept per PAIR of columns (an LRU of mdb_col_memo sets): queries that alternate over
    several table pairs find their verdicts again - no key sample (and no host sync for it) after each pair's first call; results
    are the oracle's throughout."""
    n = 1_500_000
    pairs = []
    for s in range(3):
        a = dev.gen_keys(n, 0, n, 100 + s, 0)
        b = dev.gen_keys(n, 0, n, 200 + s, n // (4 << s))
        exp = orc.join_group_count(orc.gen_keys(n, 0, n, 100 + s, 0), None, orc.gen_keys(n, 0, n, 200 + s, n // (4 << s)), None)
        pairs.append((a, b, exp))

    def run_all():
        for a, b, (ek, ec, ef, ej) in pairs:
            k, c, f, j = dev.join_group_count(a, None, b, None)
            assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec)
    run_all()                   # every pair's first call: samples
    dev.prof_enable(True)
    dev.prof_reset()
    for _ in range(3):
        run_all()
    prof = dev.prof_read()
    dev.prof_enable(False)
    assert prof.get("key_sample", (0, 0.0))[0] == 0, prof.get("key_sample")
    # a column written through the library (upload) loses what was known about it - and only it
    host = _np(pairs[0][0])
    dev._chk(dev.lib.mdb_dev_h2d(dev.h, pairs[0][0].data_ptr(), host.ctypes.data, 8 * n), "h2d")
    dev.prof_enable(True)
    dev.prof_reset()
    run_all()
    prof = dev.prof_read()
    dev.prof_enable(False)
    assert prof.get("key_sample", (0, 0.0))[0] == 1, prof.get("key_sample")


@pytest.mark.parametrize("shape", ["unique_3", "dups_3", "four_tables", "selective", "wide_keys", "small", "skew", "nulls"])
def test_join_group_count_multi_same_key(dev, shape):
    """mdb_dev_join_group_count_multi: L JOIN R0 JOIN R1 [JOIN R2] on ONE key + GROUP BY + COUNT(*) (BASELINE configs[4] shape) - every table
    partitioned once, the right tables' counts multiplied in the leaf kernel; keys that do not take the compact form, skew, tiny tables go
    through the chain of two-table operators inside the same call.  Oracle: per-key products of the tables' counts, in first-occurrence order."""
    rng = np.random.default_rng(len(shape) * 7)
    n = {"small": 3000}.get(shape, 1_300_000)
    span = {"unique_3": n, "dups_3": n // 8, "four_tables": n // 3, "selective": n, "wide_keys": n, "small": 500, "skew": n // 4, "nulls": n // 2}[shape]
    off = 10**14 if shape == "wide_keys" else 1000
    mk = lambda m, s=span: off + rng.integers(0, s, m, dtype=np.int64) * (2**33 if shape == "wide_keys" else 1)  # noqa: E731
    kl = off + rng.permutation(n).astype(np.int64) if shape == "unique_3" else mk(n)
    rights = [mk(n), mk(n - 1000)] + ([mk(n // 2)] if shape == "four_tables" else [])
    if shape == "selective":
        rights[0] = off + rng.integers(0, n // 20, n // 4, dtype=np.int64)
    if shape == "skew":
        rights[1][rng.random(len(rights[1])) < 0.6] = off + 77
        kl[rng.random(n) < 0.3] = off + 77
    nl = (rng.random(n) < 0.02) if shape == "nulls" else None
    nrs = [(rng.random(len(r)) < 0.02) if shape == "nulls" else None for r in rights]
    # oracle: the two-table oracle chained (counts multiplied through first-occurrence order)
    ek, ec, ef, _ = orc.join_group_count(kl, nl, rights[0], nrs[0])
    for r, nr_ in zip(rights[1:], nrs[1:]):
        k2, c2, f2, _ = orc.join_group_count(ek, None, r, nr_)
        ec, ef, ek = ec[f2] * c2, ef[f2], k2
    k, c, f, j = dev.join_group_count_multi(dev.to_dev(kl), dev.nullbits_dev(nl), [(dev.to_dev(r), dev.nullbits_dev(m)) for r, m in zip(rights, nrs)])
    assert np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec)
    assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef)
    assert j == int(ec.sum())
    if shape != "nulls":
        assert dev.last_join_multi() == (shape in ("unique_3", "dups_3", "four_tables", "selective")), shape
    # the same call without MDB_ORDER_FIRST and without first rows: the same groups in any order
    k, c, j = dev.join_group_count_multi_unordered(dev.to_dev(kl), dev.nullbits_dev(nl), [(dev.to_dev(r), dev.nullbits_dev(m)) for r, m in zip(rights, nrs)])
    assert j == int(ec.sum()) and dict(zip(_np(k).tolist(), _np(c).tolist())) == dict(zip(ek.tolist(), ec.tolist())) and k.numel() == len(ek)
    if shape in ("unique_3", "dups_3", "four_tables"):      # ("selective": under 2^21 rows in the first two tables - the ordered operator answers)
        assert dev.last_join_unordered(), shape


@pytest.mark.parametrize("shape", ["dense_unique", "dim_in_low_range", "dups_spread", "nulls_both", "window_far_from_zero", "keys_beyond_any_window",
                                   "skew", "small", "sparse_keys_window_2^25", "sparse_keys_window_2^27_nulls_dups", "window_2^24"])
def test_join_group_count_without_order(dev, shape):
    """mdb_dev_join_group_count without MDB_ORDER_FIRST and without first rows: the groups may come in any order, and the operator then
    moves no row ids and sorts nothing (the sharded operator's receiver pipeline on this GPU's own regions).  Same groups and counts as
    the oracle, as a set; shapes the form does not serve (keys beyond any 2^30-value window, skew, small tables) are answered by the
    ordered operator through the same call."""
    rng = np.random.default_rng(len(shape))
    n = 3000 if shape == "small" else 2_400_000
    off = {"window_far_from_zero": -(2**50), "keys_beyond_any_window": 0}.get(shape, 77)
    kl = off + rng.permutation(n).astype(np.int64)
    kr = off + rng.permutation(n).astype(np.int64)[: n - 999]
    if shape == "dim_in_low_range":
        kr = off + rng.integers(0, n // 16, n, dtype=np.int64)
    elif shape == "dups_spread":
        kr = off + 16 * rng.integers(0, n // 16, n, dtype=np.int64)
        kl = off + rng.integers(0, n, n, dtype=np.int64)
    elif shape == "keys_beyond_any_window":
        kl = rng.integers(-2**62, 2**62, n, dtype=np.int64)
        kr = np.concatenate([kl[: n // 2], rng.integers(-2**62, 2**62, n // 2, dtype=np.int64)])
    elif shape == "sparse_keys_window_2^25":     # (windows of 2^24 .. 2^27 values: the 4096-digit first level, one pass per table)
        kl, kr = off + 8 * (kl - off), off + 8 * (kr - off)
    elif shape == "sparse_keys_window_2^27_nulls_dups":
        kl = off + 40 * rng.integers(0, n, n, dtype=np.int64)
        kr = off + 40 * rng.integers(0, n, n + 12345, dtype=np.int64)
    elif shape == "window_2^24":
        kl, kr = off - 2**40 + 5 * (kl - off), off - 2**40 + 5 * (kr - off)
    elif shape == "skew":
        kl[rng.random(n) < 0.7] = off + 5
        kr[rng.random(len(kr)) < 0.5] = off + 5
    nl = (rng.random(n) < 0.02) if "nulls" in shape else None
    nr = (rng.random(len(kr)) < 0.02) if "nulls" in shape else None
    ek, ec, _, ej = orc.join_group_count(kl, nl, kr, nr)
    k, c, j = dev.join_group_count_unordered(dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr), dev.nullbits_dev(nr))
    got = dict(zip(_np(k).tolist(), _np(c).tolist()))
    assert len(got) == k.numel() and got == dict(zip(ek.tolist(), ec.tolist())) and j == ej
    assert dev.last_join_unordered() == (shape not in ("keys_beyond_any_window", "skew", "small")), shape


@pytest.mark.parametrize("shape", ["dups16", "unique_window_2^22", "sparse_window_2^26", "far_from_zero", "few_values", "beyond_any_window", "nulls", "small", "skew"])
def test_group_count_keys_in_any_order(dev, shape):
    """mdb_dev_group_count_keys: GROUP BY + COUNT(*) of one column as (key, COUNT) pairs in unspecified order - the same groups as the
    oracle's, as a set; what the form does not serve (NULL keys, keys beyond any 2^30-value window, skew, small tables) returns 1 and
    the ordered operator answers."""
    rng = np.random.default_rng(len(shape) + 3)
    n = 3000 if shape == "small" else 2_500_000
    k = {"dups16": lambda: 9 + rng.integers(0, n // 16, n, dtype=np.int64),
         "unique_window_2^22": lambda: -5 + rng.permutation(n).astype(np.int64),
         "sparse_window_2^26": lambda: 7 + 20 * rng.integers(0, n, n, dtype=np.int64),
         "far_from_zero": lambda: 2**55 + rng.integers(0, n // 3, n, dtype=np.int64),
         "few_values": lambda: rng.integers(0, 100, n, dtype=np.int64),
         "beyond_any_window": lambda: rng.integers(-2**62, 2**62, n, dtype=np.int64),
         "nulls": lambda: rng.integers(0, n // 4, n, dtype=np.int64),
         "small": lambda: rng.integers(0, 50, n, dtype=np.int64),
         "skew": lambda: np.where(rng.random(n) < 0.8, 12345, rng.integers(0, n, n, dtype=np.int64))}[shape]()
    nulls = (rng.random(n) < 0.01) if shape == "nulls" else None
    got = dev.group_count_keys(dev.to_dev(k), dev.nullbits_dev(nulls))
    if shape in ("beyond_any_window", "nulls", "small", "skew", "few_values"):
        if shape != "few_values":
            assert got is None, shape
        if got is None:
            return
    assert got is not None, shape
    keys, counts = got
    vals, cnt = np.unique(k, return_counts=True)
    res = dict(zip(_np(keys).tolist(), _np(counts).tolist()))
    assert len(res) == keys.numel() and res == dict(zip(vals.tolist(), cnt.tolist()))
    ef, ec = orc.group_count(k, None)
    assert sorted(res.items()) == sorted(zip(k[ef].tolist(), ec.tolist()))


@pytest.mark.parametrize("seed", list(range(12)))
def test_any_order_operators_random_shapes(dev, seed):
    """Random table sizes (odd, not multiples of any tile), key windows of 2^13 ... 2^29 values anywhere in the int64 range, duplication on
    either side, NULLs: the any-order join + GROUP BY (one partition level of 512 or 4096 digits, or two levels) and the any-order GROUP BY
    against the oracle, as sets."""
    rng = np.random.default_rng(1000 + seed)
    n_l = int(rng.integers(1_100_000, 3_500_000)) | 1
    n_r = int(rng.integers(1_100_000, 3_500_000))
    kbits = int(rng.integers(13, 30))
    span = int(2 ** kbits * rng.uniform(0.55, 0.85))
    base = int(rng.integers(-2**60, 2**60))
    kl = base + rng.integers(0, span, n_l, dtype=np.int64)
    kr = base + rng.integers(0, max(span // int(rng.integers(1, 20)), 1), n_r, dtype=np.int64)
    nl = (rng.random(n_l) < 0.03) if seed % 3 == 0 else None
    nr = (rng.random(n_r) < 0.03) if seed % 3 == 1 else None
    ek, ec, _, ej = orc.join_group_count(kl, nl, kr, nr)
    k, c, j = dev.join_group_count_unordered(dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr), dev.nullbits_dev(nr))
    assert j == ej and k.numel() == len(ek)
    assert dict(zip(_np(k).tolist(), _np(c).tolist())) == dict(zip(ek.tolist(), ec.tolist())), (seed, kbits, n_l, n_r)
    got = dev.group_count_keys(dev.to_dev(kr), None)
    if got is not None:
        vals, cnt = np.unique(kr, return_counts=True)
        assert dict(zip(_np(got[0]).tolist(), _np(got[1]).tolist())) == dict(zip(vals.tolist(), cnt.tolist())), (seed, kbits)


def test_join_pairs_reports_an_identity_left_vector(dev):
    """mdb_dev_last_pairs_identity: unique right keys and a partner for every left row (the primary-key join of BASELINE configs[1])
    -> the left vector is 0, 1, 2 ... and says so; one left row without a partner, or a right key twice, and it does not."""
    rng = np.random.default_rng(5)
    n = 700_000
    a = rng.permutation(n).astype(np.int64) + 10
    b = rng.permutation(n).astype(np.int64) + 10
    l, r = dev.join_pairs(dev.to_dev(a), None, dev.to_dev(b), None)
    assert dev.last_pairs_identity() and np.array_equal(_np(l).view(np.uint32), np.arange(n, dtype=np.uint32))
    a2 = a.copy()
    a2[123] = -5                                        # a left row without a partner
    l, r = dev.join_pairs(dev.to_dev(a2), None, dev.to_dev(b), None)
    assert not dev.last_pairs_identity() and l.numel() == n - 1
    b2 = np.concatenate([b, b[:1]])                     # a right key twice: n + 1 pairs
    l, r = dev.join_pairs(dev.to_dev(a), None, dev.to_dev(b2), None)
    assert not dev.last_pairs_identity() and l.numel() == n + 1


@pytest.mark.parametrize("shape", ["unique_both", "unique_both_nulls", "right_duplicates", "left_duplicates", "window_2^26"])
def test_join_keys_in_the_references_order_for_primary_key_joins(dev, shape):
    """mdb_dev_join_keys_ordered: a join whose only output is its key column, in the reference's left-major order
    (executor_select.c:1096-1141), answered by the ordered join + GROUP BY + COUNT(*) operator when every key has one row on either side
    (J == G), or one row in the LEFT table (the direct-address leaf kernels say so): every key COUNT times at its left row's place - the keys
    of the oracle's pairs; duplicate LEFT keys are left to mdb_dev_join_pairs (None), remembered."""
    rng = np.random.default_rng(len(shape) * 3)
    n_l, n_r = 2_300_000 + 1000 * len(shape), 2_100_000 + 777 * len(shape)      # (what the operator remembers about a column pair goes by address and length)
    span = (1 << 26) - 5 if shape == "window_2^26" else 3_000_000
    kl = rng.permutation(span)[:n_l].astype(np.int64) - 1000
    kr = rng.permutation(span)[:n_r].astype(np.int64) - 1000
    nl = nr = None
    if shape == "unique_both_nulls":
        nl, nr = rng.random(n_l) < 0.05, rng.random(n_r) < 0.05
    if shape == "right_duplicates":
        kr[:1000] = kr[1000:2000]
        kr[5] = kl[7]
        kr[6] = kl[7]
    if shape == "left_duplicates":
        kl[10:20] = kl[30]
        kr[3] = kl[30]
    el, er = orc.join_pairs(kl, nl, kr, nr)
    dl, dr, dnl, dnr = dev.to_dev(kl), dev.to_dev(kr), dev.nullbits_dev(nl), dev.nullbits_dev(nr)
    for round_ in range(2):
        got = dev.join_keys_ordered(dl, dnl, dr, dnr)
        if shape == "left_duplicates":       # (the materialising join answers: rows of one key are not adjacent in the left-major order)
            assert got is None, (shape, round_)
        else:                               # (duplicates on the right side only: every key COUNT times at its left row's place)
            assert got is not None and np.array_equal(_np(got), kl[el]), (shape, round_)


@pytest.mark.parametrize("seed", range(10))
def test_join_keys_in_the_references_order_random_shapes(dev, seed):
    """Random sizes, key windows (one 9-bit level, two levels, the 4096-digit pass, keys outside every window), right-side multiplicities
    from 1 to 70, NULLs: whenever mdb_dev_join_keys_ordered answers, it is the key column of the oracle's pairs in their order; duplicate
    left keys among the matched ones always make it decline."""
    rng = np.random.default_rng(4000 + seed)
    n_l = int(rng.integers(1_100_000, 2_400_000)) + seed
    n_r = int(rng.integers(1_100_000, 2_400_000)) + 3 * seed
    bits = int(rng.choice([21, 22, 23, 25, 26, 40]))
    span = (1 << bits) - int(rng.integers(1, 1 << (bits - 3))) if bits < 40 else None
    kl = (rng.permutation(span)[:n_l] if span else rng.integers(0, 1 << 40, n_l)).astype(np.int64)
    n_l = len(kl)
    if span is None:
        kl = np.unique(kl)
        rng.shuffle(kl)
        n_l = len(kl)
    mult = int(rng.choice([1, 1, 2, 7, 30, 70]))
    pool = kl[rng.integers(0, n_l, max(n_r // mult, 1))]
    kr = pool[rng.integers(0, len(pool), n_r)].astype(np.int64)
    left_dups = rng.random() < 0.3
    if left_dups:
        kl[100:103] = kr[17]
    nl = nr = None
    if rng.random() < 0.4:
        nl, nr = rng.random(n_l) < 0.03, rng.random(n_r) < 0.03
        if left_dups:
            nl[100:103] = False
            nr[17] = False
    el, er = orc.join_pairs(kl, nl, kr, nr)
    got = dev.join_keys_ordered(dev.to_dev(kl), dev.nullbits_dev(nl), dev.to_dev(kr), dev.nullbits_dev(nr))
    if left_dups:
        assert got is None
    elif got is not None:
        assert np.array_equal(_np(got), kl[el]), (seed, bits, mult)


# ---- round 5: catalog statistics instead of key samples (mdb_dev_call_stats), and what an operator did (mdb_dev_last_plan)
@pytest.mark.parametrize("variant", ["D", "U", "S"])
def test_call_stats_replace_the_key_sample_and_the_plan_says_so(dev, variant):
    """the same join + GROUP BY with and without the caller's statistics: identical results; with them no sampling kernel runs, nothing
    is retried, and the plan is the one the sampled call arrives at"""
    n = 3_000_000
    a = dev.gen_keys(n, 0, n, 42, 0)
    b = dev.gen_keys(n, 0, n, 43, 0 if variant == "U" else n // 16)
    if variant == "S":
        b.mul_(16)
    k0, c0, f0, j0 = dev.join_group_count(a, None, b, None)
    sampled = dev.last_plan()
    assert sampled["from_stats"] == 0      # (it may have been retried: a verdict remembered by address for a buffer that now holds other data)
    a2, b2 = a.clone(), b.clone()       # columns the context has never seen: nothing remembered by address
    dev.call_stats(a2, dev.key_range(a2), b2, dev.key_range(b2))
    try:
        k1, c1, f1, j1 = dev.join_group_count(a2, None, b2, None)
        plan = dev.last_plan()
    finally:
        dev.call_stats()
    assert plan["from_stats"] == 1 and plan["samples"] == 0 and plan["retries"] == 0, plan
    if sampled["retries"] == 0:
        assert {k: plan[k] for k in ("key_form", "levels", "digits")} == {k: sampled[k] for k in ("key_form", "levels", "digits")}, (plan, sampled)
    assert j1 == j0 and torch.equal(k1, k0) and torch.equal(c1, c0) and torch.equal(f1, f0)
    # the statistics gone: a fresh pair of columns is sampled again
    a3, b3 = a.clone(), b.clone()
    dev.join_group_count(a3, None, b3, None)
    assert dev.last_plan()["from_stats"] == 0      # (sampled - or remembered by address, when the allocator handed out a buffer seen before)


def test_call_stats_that_do_not_hold_cost_a_retry_never_a_result(dev):
    """statistics are a promise the device checks: a range narrower than the column's sends the operator to the forms that need none -
    the result is the oracle's"""
    rng = np.random.default_rng(9)
    n = 1_500_000
    kl, kr = rng.integers(0, 4_000_000, n).astype(np.int64), rng.integers(0, 4_000_000, n).astype(np.int64)
    ek, ec, ef, ej = orc.join_group_count(kl, None, kr, None)
    a, b = dev.to_dev(kl), dev.to_dev(kr)
    dev.call_stats(a, (1000, 2_000_000), b, (0, 1_000_000))      # both too narrow
    try:
        k, c, f, j = dev.join_group_count(a, None, b, None)
        plan = dev.last_plan()
    finally:
        dev.call_stats()
    assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec)
    assert plan["from_stats"] == 1 and plan["retries"] >= 1, plan


def test_counters_say_what_a_call_paid_for(dev):
    """mdb_dev_counters(): running totals a caller reads around a timed loop (bench.py: retries_in_timed_steps) - a first call over new
    columns pays a key sample, statistics that do not hold pay a retry, repeated calls pay nothing"""
    n = 1_200_000
    a = dev.gen_keys(n, 0, n, 42, 0)
    b = dev.gen_keys(n, 0, n, 43, n // 16)
    c0 = dev.counters()
    dev.join_group_count(a, None, b, None)
    c1 = dev.counters()
    assert c1["operator_calls"] == c0["operator_calls"] + 1 and c1["samples"] >= c0["samples"] + 1, (c0, c1)
    for _ in range(5):
        dev.join_group_count(a, None, b, None)
    c2 = dev.counters()
    assert c2["operator_calls"] == c1["operator_calls"] + 5
    assert {k: c2[k] - c1[k] for k in ("retries", "samples", "arena_grows", "alloc_misses")} == {"retries": 0, "samples": 0, "arena_grows": 0, "alloc_misses": 0} or \
        c2["alloc_misses"] > c1["alloc_misses"], (c1, c2)      # (result columns are allocated per call: the allocator's cache may miss; nothing else moves)
    dev.call_stats(a, (1000, 2000), b, (0, 10))     # a promise the device finds broken: the operator is redone
    try:
        dev.join_group_count(a, None, b, None)
    finally:
        dev.call_stats()
    c3 = dev.counters()
    assert c3["retries"] >= c2["retries"] + 1, (c2, c3)


def test_distinct_statistic_runs_the_bit_form_without_a_pilot_and_a_wrong_flag_never_costs_a_result(dev):
    """round 6: MDB_COL_DISTINCT in mdb_dev_call_stats (what the store measures at ingest with mdb_dev_distinct_scan): two key columns without a
    value twice, the right one holding every value of its range - the join + GROUP BY leaves its groups as one bit per left row on the first
    call (groups_as_bits == 3: by statistics; no pilot, no sample); a flag that does not hold is caught by the leaf's own counting"""
    rng = np.random.default_rng(61)
    n = 17_000_000
    ka, kb = rng.permutation(n).astype(np.int64) + 40, rng.permutation(n).astype(np.int64) + 40
    a, b = dev.to_dev(ka), dev.to_dev(kb)
    assert dev.distinct(a) and dev.distinct(b)
    dev.call_stats(a, (40, n + 39, 1), b, (40, n + 39, 1))
    try:
        k, c, f, j = dev.join_group_count(a, None, b, None)
        plan = dev.last_plan()
    finally:
        dev.call_stats()
    assert plan["groups_as_bits"] == 3 and plan["samples"] == 0 and plan["retries"] == 0 and plan["digits"] == 4096, plan
    assert j == n and np.array_equal(_np(k), ka) and bool((_np(c) == 1).all()) and np.array_equal(_np(f), np.arange(n))
    # the same promise over a column that DOES hold a value twice (a raw caller's mistake): one left row finds no partner, one finds two
    kb2 = kb.copy()
    kb2[5] = kb2[6]
    b2 = dev.to_dev(kb2)
    assert not dev.distinct(b2)
    a2 = a.clone()      # (fresh addresses: nothing remembered)
    dev.call_stats(a2, (40, n + 39, 1), b2, (40, n + 39, 1))
    try:
        k, c, f, j = dev.join_group_count(a2, None, b2, None)
    finally:
        dev.call_stats()
    cnt = np.bincount(kb2 - 40, minlength=n)
    sel = cnt[ka - 40] > 0
    assert j == n and np.array_equal(_np(k), ka[sel]) and np.array_equal(_np(c), cnt[ka - 40][sel]) and np.array_equal(_np(f), np.nonzero(sel)[0])
    # GROUP BY over a distinct column: the identity (group_form 3)
    dev.call_stats(a, (40, n + 39, 1))
    try:
        gf, gc_ = dev.group_count(a, None)[:2]
        plan = dev.last_plan()
    finally:
        dev.call_stats()
    assert plan["group_form"] == 3, plan
    assert np.array_equal(_np(gf), np.arange(n)) and bool((_np(gc_) == 1).all())


def test_join_group_count_writes_no_copy_of_what_the_left_column_already_says(dev):
    """round 6: MDB_KEYS_MAY_ALIAS | MDB_COUNTS_OPTIONAL (how query_execute() calls the operator) - when every left row turns out to be a group, in
    row order, the group keys ARE the left key column: the operator writes none and says so; when every COUNT is 1 it writes no COUNT column;
    a key with two right rows keeps the COUNT column, a left row without partner keeps the key column.  Same results as the copying call."""
    rng = np.random.default_rng(62)
    n = 17_000_000
    ka, kb = rng.permutation(n).astype(np.int64) + 40, rng.permutation(n).astype(np.int64) + 40
    a, b = dev.to_dev(ka), dev.to_dev(kb)
    k0, c0, f0, j0 = dev.join_group_count(a, None, b, None, want_first=False)
    assert dev.last_plan()["digits"] == 4096
    for _ in range(2):      # (the second call runs the bit-per-row form on what the first one delivered)
        k, c, f, j = dev.join_group_count(a, None, b, None, want_first=False, no_copies=True)
    p = dev.last_plan()
    assert p["groups_as_bits"] in (1, 2) and p["keys_are_left_column"] == 1 and p["counts_all_one"] == 1, p
    assert j == n and k.data_ptr() == a.data_ptr() and k.numel() == n and c is None and torch.equal(k, k0) and bool((c0 == 1).all())
    # ... and mdb_dev_join_keys_ordered hands the left column back as the joined rows' key column
    jk = dev.join_keys_ordered(a, None, b, None)
    assert jk is not None and jk.data_ptr() == a.data_ptr() and jk.numel() == n
    # one right key twice: the COUNT column is written (its 2 at the key's place), the keys of the left rows - one of them lost its partner - too
    kb2 = kb.copy()
    kb2[5] = kb2[6]
    b2, a2 = dev.to_dev(kb2), a.clone()
    for _ in range(2):
        k, c, f, j = dev.join_group_count(a2, None, b2, None, want_first=False, no_copies=True)
    p = dev.last_plan()
    assert p["keys_are_left_column"] == 0 and p["counts_all_one"] == 0, p
    cnt = np.bincount(kb2 - 40, minlength=n)
    sel = cnt[ka - 40] > 0
    assert j == n and np.array_equal(_np(k), ka[sel]) and np.array_equal(_np(c), cnt[ka - 40][sel])
    jk = dev.join_keys_ordered(a2, None, b2, None)
    assert jk is not None and jk.data_ptr() != a2.data_ptr() and np.array_equal(_np(jk), np.repeat(ka[sel], cnt[ka - 40][sel]))


def test_join_payload_plan_names_the_form(dev, monkeypatch):
    rng = np.random.default_rng(10)
    kr = np.unique(rng.integers(0, 1 << 26, 2_000_000, dtype=np.int64))
    kl = kr[rng.integers(0, len(kr), 2_400_000)]
    pay = [rng.integers(0, 1 << 40, len(kr), dtype=np.int64)]
    for knob, form in (("2", 3), ("0", 2)):
        monkeypatch.setenv("MDB_ROWJOIN", knob)
        got = dev.join_payload(dev.to_dev(kl), None, dev.to_dev(kr), None, [dev.to_dev(p) for p in pay])
        assert got is not None and dev.last_plan()["payload_form"] == form, (knob, dev.last_plan())


def test_join_payload_older_forms_after_a_region_overflow_write_inside_their_arrays(dev, monkeypatch):
    """The sequence the payload soak faulted on (round 5): a served two-level join, then - older forms, same sizes - a join whose hot key
    (150 000 left rows of one key) overflows a fixed-capacity region.  The region's slots nobody wrote still hold row ids of the EARLIER
    call's larger table: a leaf that indexed a result column with them wrote outside it (a GPU memory access fault ends the process,
    so surviving the sequence is the assertion).  Now: not served, or served right."""
    monkeypatch.setenv("MDB_ROWJOIN", "0")
    rng = np.random.default_rng(70_014)
    span = 1 << 25
    kr = np.unique(rng.integers(0, span, 1_500_000)).astype(np.int64) + 10**15
    big_l = kr[rng.integers(0, len(kr), 3_100_001)]
    pay = [rng.integers(-2**62, 2**62, len(kr), dtype=np.int64), rng.standard_normal(len(kr))]
    got = dev.join_payload(dev.to_dev(big_l), None, dev.to_dev(kr), None, [dev.to_dev(p) for p in pay])
    assert got is not None
    order = np.argsort(kr, kind="stable")
    assert np.array_equal(_np(got[0]), pay[0][order][np.searchsorted(kr[order], big_l)])
    del got
    kl = kr[rng.integers(0, len(kr), 2_345_679)]
    kl[1_000_000:1_150_000] = kr[12345]
    for _ in range(3):
        got = dev.join_payload(dev.to_dev(kl), None, dev.to_dev(kr), None, [dev.to_dev(p) for p in pay])
        if got is not None:
            pos = np.searchsorted(kr[order], kl)
            assert all(np.array_equal(_np(g).view(np.int64), p[order][pos].view(np.int64)) for g, p in zip(got, pay))
    torch.cuda.synchronize()


@pytest.mark.parametrize("cells", [1, 2])
def test_join_payload_1e8_rows_every_left_row_gets_its_partners_cells(dev, cells):
    """BASELINE configs[4]'s join-only shape at its full size (10^8 x 10^8 unique keys in two different orders), through the row-order
    form: a size-independent property instead of an oracle run - the right table's payload cells are FUNCTIONS of its key (an INT64
    one and a DOUBLE one), so out[c][i] must equal f_c(key_l[i]) for every left row, bit for bit; and the plan says which form ran"""
    n = 100_000_000
    a = dev.gen_keys(n, 0, n, 42, 0)
    b = dev.gen_keys(n, 0, n, 43, 0)
    pay = [b * 3 + 1, (b.to(torch.float64) * 0.5 - 7.25)][:cells]
    got = dev.join_payload(a, None, b, None, pay)
    assert got is not None and dev.last_plan()["payload_form"] == 3, dev.last_plan()
    assert torch.equal(got[0], a * 3 + 1)
    if cells == 2:
        assert torch.equal(got[1].view(torch.int64), (a.to(torch.float64) * 0.5 - 7.25).view(torch.int64))
    del got, pay
