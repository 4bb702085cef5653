"""The ordered join + GROUP BY + COUNT(*) over key windows of 2^24 ... 2^27 values (10^8 unique keys per table: BASELINE
configs[2]'s unfavourable variants): ONE 4096-digit pass per table - 2-byte words for the right table, 4-byte ROW words with
run headers for the left one (mdb_dev_shard.hip: k_shard_scatter_wide<.., ROWS>) - and k_leaf_wide over digits of up to 2^15
values, two workgroups per digit.  Groups, counts, first rows and order are the oracle's
(/root/reference/src/engine/executor_select.c:1076-1149 join, 1526-1588 first-occurrence order)."""
import os

import numpy as np
import pytest
import torch

from oracle import np_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from midoridb_amd.dev import DeviceCtx
    d = DeviceCtx(0)
    yield d
    d.close()


@pytest.fixture()
def forced():
    """the form is meant for tables of 2^26 rows and more: let it run on the test's small ones"""
    os.environ["MDB_WIDE12_MIN"] = "1"
    yield
    del os.environ["MDB_WIDE12_MIN"]


def _np(t):
    return t.cpu().numpy()


def _check(dev, kl, nl, kr, nr, expect_form=True, rounds=2):
    ek, ec, ef, ej = orc.join_group_count(kl, nl, kr, nr)
    dl, dr, dnl, dnr = dev.to_dev(kl), dev.to_dev(kr), dev.nullbits_dev(nl), dev.nullbits_dev(nr)
    for round_ in range(rounds):     # the second call runs on remembered verdicts (4-byte records among them)
        k, c, f, j = dev.join_group_count(dl, dnl, dr, dnr)
        assert j == ej, (round_, j, ej)
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef), round_
        assert np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec), round_
        if expect_form is not None:
            assert dev.last_join_one_pass_4096() == expect_form, (round_, dev.last_join_form())


@pytest.mark.parametrize("bits", [24, 25, 26, 27])
@pytest.mark.parametrize("shape", ["unique", "dup16_right", "n_to_m", "nulls", "negative_offset", "offset_1e12"])
def test_one_pass_4096_matches_the_oracle(dev, forced, bits, shape):
    rng = np.random.default_rng(bits * 31 + len(shape))
    span = (1 << bits) - int(rng.integers(1, 1 << (bits - 2)))      # the window rounds up to 2^bits
    n_l, n_r = 1_200_000 + int(rng.integers(0, 5000)), 1_000_000 + int(rng.integers(0, 5000))
    off = {"negative_offset": -(2**41) - 12345, "offset_1e12": 10**12 + 7}.get(shape, 0)
    if shape == "n_to_m":
        kl = off + rng.integers(0, span, n_l, dtype=np.int64)
        kr = off + rng.integers(0, span, n_r, dtype=np.int64)
        kl[: n_l // 2] = kl[n_l // 2: 2 * (n_l // 2)]               # every key of the first half twice on the left
        kr[: n_r // 3] = kl[: n_r // 3]                             # ... and a third of the right rows among them
    else:
        pl = rng.permutation(span)[: max(n_l, n_r)].astype(np.int64)
        kl = off + pl[:n_l]
        pr = rng.permutation(span)[:n_r].astype(np.int64)
        kr = off + (pr if shape != "dup16_right" else 16 * (pr % (span // 16)))
    # the sample must see the window's ends (2 x 4096 pseudo-random positions of 10^6 rows do not): pin them
    kl[0], kl[-1] = off, off + span - 1
    nl = nr = None
    if shape == "nulls":
        nl, nr = rng.random(n_l) < 0.07, rng.random(n_r) < 0.15
        nl[0] = nl[-1] = False
        kl[nl] = rng.integers(-2**62, 2**62, int(nl.sum()))
        kr[nr] = rng.integers(-2**62, 2**62, int(nr.sum()))
    _check(dev, kl, nl, kr, nr, expect_form=None)
    assert dev.last_join_form() in (1, 2)
    # whether the compact window was offered is the sample's business; when it was, the 4096-digit pass must have run
    if dev.last_join_form() == 2:
        assert dev.last_join_one_pass_4096()


def test_one_pass_4096_a_key_with_70000_rows_falls_back(dev, forced):
    """16-bit row counts per key value: a key with more rows is reported by the leaf kernel and the operator redone with two
    levels (and their hot-key path) - same result."""
    rng = np.random.default_rng(5)
    span, n_l, n_r = (1 << 26) - 77, 1_500_000, 1_500_000
    kl = rng.permutation(span)[:n_l].astype(np.int64)
    kr = rng.permutation(span)[:n_r].astype(np.int64)
    kl[0], kl[-1] = 0, span - 1
    kr[200_000:270_000] = 4242
    kl[1000:1003] = 4242
    _check(dev, kl, None, kr, None, expect_form=None, rounds=2)
    assert not dev.last_join_one_pass_4096()


@pytest.mark.parametrize("side,rows", [("right", 31), ("right", 32), ("right", 40), ("left", 15), ("left", 16), ("left", 20), ("both", 15)])
def test_one_pass_4096_count_fields_at_and_beyond_their_limits(dev, forced, side, rows):
    """5 bits of right rows and 4 bits of left rows per key value in the leaf kernel's table: 31 / 15 rows fit, one more is noticed (the sums of
    the fields fall short of the rows counted) and answered by the two-level form - the oracle's result either way."""
    rng = np.random.default_rng(rows * 3 + len(side))
    # (sizes of their own per case: what the operator remembers about a column pair goes by address and length, and the allocator hands the same
    # addresses out again)
    span, n_l, n_r = (1 << 25) - 4321, 1_100_000 + 1000 * rows + 37 * len(side), 1_050_000 + 512 * rows
    kl = rng.permutation(span)[:n_l].astype(np.int64)
    kr = rng.permutation(span)[:n_r].astype(np.int64)
    kl[0], kl[-1] = 0, span - 1
    key = int(kl[5])
    kr[kr == key] = key + 1 if key + 1 < span else key - 1
    if side in ("right", "both"):
        kr[rng.choice(n_r, rows if side == "right" else 31, replace=False)] = key
    else:
        kr[77] = key
    if side in ("left", "both"):
        idx = rng.choice(np.arange(10, n_l - 10), rows - 1, replace=False)
        kl[idx] = key
    _check(dev, kl, None, kr, None, expect_form=None)
    fits = (side == "right" and rows <= 31) or (side == "left" and rows <= 15) or side == "both"
    assert dev.last_join_one_pass_4096() == fits, (side, rows)


def test_one_pass_4096_left_rows_outside_the_right_tables_window(dev, forced):
    """the right table's keys fill a 2^26 window, the left table's spread over 2^31: the window is the right table's (by_span), left rows
    outside it have no partner and are dropped by the pass"""
    rng = np.random.default_rng(6)
    n_l, n_r = 3_000_000, 1_500_000
    base = 5 * 2**26 + 1234
    kr = base + rng.permutation((1 << 26) - 999)[:n_r].astype(np.int64)
    kl = rng.integers(0, 1 << 31, n_l, dtype=np.int64)
    kl[::7] = kr[rng.integers(0, n_r, len(kl[::7]))]
    _check(dev, kl, None, kr, None, expect_form=None)


def test_one_pass_4096_at_scale_equals_the_two_level_form(dev):
    """4 * 10^7 x 4 * 10^7 rows of the benchmark's generator, variants U and S: the one-pass form (default from 2^26 rows in all) and
    the two-level form (MDB_WIDE12=0) deliver identical columns."""
    n = 40_000_000
    kl = dev.gen_keys(n, 0, n, 42, 0)
    for variant in ("U", "S"):
        kr = dev.gen_keys(n, 0, n, 43, 0 if variant == "U" else n // 16)
        if variant == "S":
            kr = kr * 16
        out = {}
        for mode in ("1", "0"):
            os.environ["MDB_WIDE12"] = mode
            try:
                for _ in range(2):
                    k, c, f, j = dev.join_group_count(kl, None, kr, None)
                assert dev.last_join_one_pass_4096() == (mode == "1"), (variant, mode)
                out[mode] = (k.clone(), c.clone(), f.clone(), j)
            finally:
                del os.environ["MDB_WIDE12"]
        for a, b in zip(out["1"][:3], out["0"][:3]):
            assert torch.equal(a, b), variant
        assert out["1"][3] == out["0"][3]
        del out
        torch.cuda.empty_cache()


@pytest.mark.parametrize("side,rows", [("right", 16), ("right", 31), ("right", 32), ("right", 300), ("left", 15), ("left", 16), ("left", 200)])
def test_one_level_leaf_with_4_byte_table_entries_and_its_fallback(dev, side, rows):
    """Key windows of up to 2^23 values (one 9-bit level): k_leaf_wide4 keeps 5 bits of right rows and 4 bits of left rows per key value
    (two workgroups per CU); a key with more is noticed and the SAME partitioned tables go through k_leaf_wide's 16-bit counts - the
    oracle's result, one partition level, on the first call and on the remembered verdict."""
    rng = np.random.default_rng(rows * 5 + len(side))
    span, n_l, n_r = (1 << 21) - 999, 1_300_000 + 1000 * rows + 41 * len(side), 1_200_000 + 640 * rows
    kl = rng.permutation(span)[:n_l].astype(np.int64)
    kr = rng.integers(0, span, n_r, dtype=np.int64)
    kl[0], kl[-1] = 0, span - 1
    key = int(kl[9])
    kr[kr == key] = key + 1 if key + 1 < span else key - 1
    # (random right keys: a few values per key at most - far below the fields' limits - besides the planted one)
    if side == "right":
        kr[rng.choice(n_r, rows, replace=False)] = key
    else:
        kr[123] = key
        kl[rng.choice(np.arange(20, n_l - 20), rows - 1, replace=False)] = key
    _check(dev, kl, None, kr, None, expect_form=None, rounds=3)
    assert dev.last_join_form() == 2 and dev.last_join_levels() == 1


@pytest.mark.parametrize("shape", ["unique", "dups", "nulls", "product_beyond_31", "further_table_16_rows_of_a_key", "keys_outside_the_window"])
def test_one_pass_4096_three_tables_on_one_key(dev, forced, shape):
    """A JOIN B ON a = b JOIN C ON a = c GROUP BY a (BASELINE configs[4]'s shape) through the one-pass form: the further right table goes
    through the same 4096-digit pass, its rows are counted into the leaf's 4-bit fields and the two right counts multiplied before the left
    rows come.  Oracle: the two-table oracle chained; counts that do not fit the fields are answered by the two-level form."""
    rng = np.random.default_rng(len(shape) * 11)
    span = (1 << 25) - 777
    n = 1_150_000 + 97 * len(shape)
    kl = rng.permutation(span)[:n].astype(np.int64)
    kl[0], kl[-1] = 0, span - 1
    # (right tables of at most 2 and 3 rows per key: products of counts that fit the leaf's 5-bit field)
    pb, pc = rng.permutation(n), rng.permutation(n)
    rb = np.concatenate([kl[pb[: n // 2]], kl[pb[: n // 2 - 500]]])
    rc = np.concatenate([kl[pc[: n // 3]], kl[pc[: n // 3]], kl[pc[: n // 3 - 900]]])
    if shape == "unique":
        rb, rc = rng.permutation(kl)[: n - 500], rng.permutation(kl)[: n - 900]
    elif shape == "product_beyond_31":
        rb[:8], rc[:8] = kl[11], kl[11]                                             # 8 x 8 = 64 > 31
    elif shape == "further_table_16_rows_of_a_key":
        rc[:16] = kl[13]
        rb[5] = kl[13]
    elif shape == "keys_outside_the_window":
        rc = rc.copy()
        rc[::50] = span + 10**9 + rng.integers(0, 10**6, len(rc[::50]))             # dropped: they join nothing
    nl = nb = nc = None
    if shape == "nulls":
        nl, nb, nc = rng.random(n) < 0.03, rng.random(len(rb)) < 0.05, rng.random(len(rc)) < 0.08
        nl[0] = nl[-1] = False
    ek, ec, ef, _ = orc.join_group_count(kl, nl, rb, nb)
    k2, c2, f2, _ = orc.join_group_count(ek, None, rc, nc)
    ec, ef, ek = ec[f2] * c2, ef[f2], k2
    dl, db, dc = dev.to_dev(kl), dev.to_dev(rb), dev.to_dev(rc)
    for round_ in range(2):
        k, c, f, j = dev.join_group_count_multi(dl, dev.nullbits_dev(nl), [(db, dev.nullbits_dev(nb)), (dc, dev.nullbits_dev(nc))])
        assert np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec), (shape, round_)
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef)
        assert j == int(ec.sum())
    # (counts that do not fit: the two-level form, or the chain of two-table operators - whose own calls may take the one-pass form)
    if shape not in ("product_beyond_31", "further_table_16_rows_of_a_key"):
        assert dev.last_join_multi() and dev.last_join_one_pass_4096(), shape


@pytest.mark.parametrize("seed", range(int(os.environ.get("MDB_FUZZ_SEEDS_4096", "16"))))
def test_one_pass_4096_random_shapes(dev, forced, seed):
    """Random table sizes, windows of 2^23 ... 2^27 values anywhere in the int64 range, duplicate factors up to and beyond the leaf's count
    fields, NULL fractions, left rows outside the right table's range: always the oracle's groups, counts, first rows and order - through the
    one-pass form where it applies and through whatever answers when it does not."""
    rng = np.random.default_rng(1000 + seed)
    bits = int(rng.integers(23, 28))
    span = (1 << bits) - int(rng.integers(1, 1 << (bits - 3)))
    n_l, n_r = int(rng.integers(1_000_000, 2_600_000)), int(rng.integers(400_000, 2_600_000))
    off = int(rng.choice([0, 12345, -(2**45), 10**13]))
    dup_l, dup_r = int(rng.choice([1, 1, 2, 3, 9, 17])), int(rng.choice([1, 1, 2, 5, 16, 33]))
    base_l = rng.permutation(span)[: max(n_l // dup_l, 1)]
    kl = off + base_l[rng.integers(0, len(base_l), n_l)].astype(np.int64) if dup_l > 1 else off + rng.permutation(span)[:n_l].astype(np.int64)
    pool = base_l if rng.random() < 0.5 else rng.permutation(span)[: max(n_r // dup_r, 1)]
    kr = off + pool[rng.integers(0, len(pool), n_r)].astype(np.int64)
    kl[0], kl[-1] = off, off + span - 1
    if rng.random() < 0.25:          # some left rows far outside the right table's window
        kl[rng.integers(1, len(kl) - 1, len(kl) // 50)] += 2**33
    nl = nr = None
    if rng.random() < 0.4:
        nl, nr = rng.random(len(kl)) < 0.04, rng.random(n_r) < 0.06
        nl[0] = nl[-1] = False
    _check(dev, kl, nl, kr, nr, expect_form=None, rounds=2)


@pytest.mark.parametrize("shape", ["dim_in_low_range", "groups_bunched_in_the_first_rows", "more_groups_than_last_time"])
def test_one_level_leaf_writes_its_records_into_the_ordering_ranges(dev, shape):
    """From the second call over the same columns on, a one-level join whose group count is known and small enough writes its group records
    straight into the ordering kernel's ranges of 2^16 row ids (no record list, no sort levels): the oracle's result; a range with more
    groups than its region holds (groups bunched in the table's first rows; other data at the same addresses) sends the records through the
    list and its sort."""
    rng = np.random.default_rng(len(shape) * 13)
    n_l, n_r = 2_600_000 + 31 * len(shape), 2_300_000
    span = 1 << 20
    if shape == "groups_bunched_in_the_first_rows":
        # every left key that has a partner sits in the first 60 000 rows: one range of 2^16 ids gets all the groups
        kl = (span + rng.permutation(4 * n_l)[:n_l]).astype(np.int64)
        kl[:60_000] = rng.permutation(span)[:60_000]
        kr = rng.permutation(span)[:60_000][rng.integers(0, 60_000, n_r)].astype(np.int64)
        kr[:60_000] = kl[:60_000]
    else:
        kl = rng.permutation(16 * span)[:n_l].astype(np.int64)
        kr = rng.integers(0, span, n_r, dtype=np.int64)
    dl, dr = dev.to_dev(kl), dev.to_dev(kr)
    ek, ec, ef, ej = orc.join_group_count(kl, None, kr, None)
    ranged = []
    for round_ in range(3):
        if shape == "more_groups_than_last_time" and round_ == 2:
            # other data in the same buffers: many more groups than the remembered count
            kl2 = rng.permutation(span + span // 2)[:n_l].astype(np.int64) if n_l <= span + span // 2 else None
            kl = rng.integers(0, span, n_l, dtype=np.int64)
            dl.copy_(torch.from_numpy(kl).to(dl.device))
            ek, ec, ef, ej = orc.join_group_count(kl, None, kr, None)
        k, c, f, j = dev.join_group_count(dl, None, dr, None)
        assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec), (shape, round_)
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef), (shape, round_)
        ranged.append(dev.last_join_ranged_order())
    if shape == "dim_in_low_range":
        assert ranged == [False, True, True], ranged


def test_ordering_launched_before_the_group_count_is_known_stays_inside_the_callers_columns(dev):
    """With the ranges filled by the leaf kernel, the ordering kernel is launched right behind it, before the host has seen the group
    count: result columns SMALLER than the groups are reported (the operator's capacity error), and nothing is written behind them."""
    from ctypes import byref, c_uint64, c_void_p
    from midoridb_amd import dev as D
    rng = np.random.default_rng(77)
    n_l, n_r, span = 2_600_077, 2_300_000, 1 << 20
    kl = rng.permutation(16 * span)[:n_l].astype(np.int64)
    kr = rng.integers(0, span, n_r, dtype=np.int64)
    dl, dr = dev.to_dev(kl), dev.to_dev(kr)
    ek, ec, ef, ej = orc.join_group_count(kl, None, kr, None)
    G = len(ek)
    for _ in range(2):
        k, c, f, j = dev.join_group_count(dl, None, dr, None)
    assert dev.last_join_ranged_order() and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec)
    cap, guard = G - 1000, 4096
    SENT = -0x0123456789ABCDEF
    ok = torch.full((cap + guard,), SENT, dtype=torch.int64, device=dl.device)
    oc = torch.full((cap + guard,), SENT, dtype=torch.int64, device=dl.device)
    of = torch.full((cap + guard,), -7, dtype=torch.int32, device=dl.device)
    g, j = c_uint64(), c_uint64()
    rc = dev.lib.mdb_dev_join_group_count(dev.h, c_void_p(dl.data_ptr()), None, n_l, c_void_p(dr.data_ptr()), None, n_r, D.MDB_ORDER_FIRST,
                                          c_void_p(ok.data_ptr()), c_void_p(oc.data_ptr()), c_void_p(of.data_ptr()), cap, byref(g), byref(j))
    assert rc != 0 and b"capacity" in (dev.lib.mdb_dev_last_error(dev.h) or b"")
    torch.cuda.synchronize()
    assert bool((ok[cap:] == SENT).all()) and bool((oc[cap:] == SENT).all()) and bool((of[cap:] == -7).all())
    # exactly enough room: served, and the sentinel behind the last group is still there
    rc = dev.lib.mdb_dev_join_group_count(dev.h, c_void_p(dl.data_ptr()), None, n_l, c_void_p(dr.data_ptr()), None, n_r, D.MDB_ORDER_FIRST,
                                          c_void_p(ok.data_ptr()), c_void_p(oc.data_ptr()), c_void_p(of.data_ptr()), G, byref(g), byref(j))
    assert rc == 0 and g.value == G and j.value == ej
    assert np.array_equal(_np(ok[:G]), ek) and np.array_equal(_np(oc[:G]), ec) and int(ok[G]) == SENT and int(of[G]) == -7


@pytest.mark.parametrize("shape", ["unique_both", "a_few_exceptions", "many_exceptions_next_time"])
def test_one_pass_4096_nearly_every_left_row_a_group_of_count_1_leaves_as_bits(dev, forced, shape, monkeypatch):
    """When nearly every left row is a group of COUNT 1 (variant U: two primary keys) - measured by a pilot over 64 key digits on a first
    statement, known from the last call afterwards - the join clears
    one bit per left row that is NO group's first row and lists the groups whose COUNT is not 1 (k_leaf_wide12<0, true>), and
    mdb_dev_dense.hip writes key, COUNT and first row from the bits - no record per group, no ordering sort.  Same groups, counts, first
    rows and order as the oracle: left rows without partner, a left key twice, right keys twice and 20 times; and when the data behind
    the same addresses has changed to many exceptions the list overflows and the record form answers (same result)."""
    rng = np.random.default_rng(len(shape))
    n = 4_300_000 + 77 + 4096 * len(shape)      # (what the operator remembers goes by address AND length: not the previous case's columns)
    kl = rng.permutation(1 << 26)[:n].astype(np.int64) - 3_000_000_000
    kr = rng.permutation(kl)
    if shape != "unique_both":
        kr[:30_000] = kr[30_000:60_000]                 # 30 000 right keys twice (COUNT 2), 30 000 left rows lose their partner
        kr[100_000:100_020] = kr[7]                     # one right key 21 times
        kl[200_000:205_000] = kl[300_000:305_000]       # 5000 left keys twice: the later row is no first row, COUNT 2 (x right rows)
    dl, dr = dev.to_dev(kl), dev.to_dev(kr)
    ek, ec, ef, ej = orc.join_group_count(kl, None, kr, None)
    for round_ in range(3):
        if round_ == 2 and shape == "many_exceptions_next_time":
            kr2 = kr.copy()
            kr2[: n // 2] = kr2[n // 2: 2 * (n // 2)]   # half of the right keys twice: 2 x 10^6 exceptions, more than the list holds
            dr.copy_(torch.from_numpy(kr2))
            ek, ec, ef, ej = orc.join_group_count(kl, None, kr2, None)
        dev.prof_enable(True)
        dev.prof_reset()
        k, c, f, j = dev.join_group_count(dl, None, dr, None)
        ran = {kk for kk, v in dev.prof_read().items() if v[0] > 0}
        dev.prof_enable(False)
        assert j == ej and np.array_equal(_np(k), ek) and np.array_equal(_np(c), ec), (shape, round_)
        assert np.array_equal(_np(f).view(np.uint32).astype(np.int64), ef), (shape, round_)
        assert dev.last_join_one_pass_4096(), (shape, round_)
        # round 0: nothing is remembered about the columns - a pilot over 64 digits decides; later: what the last call delivered
        assert "leaf_join_wide12_bits" in ran, (shape, round_, ran)
        bits = dev.last_plan()["groups_as_bits"]      # 1: a pilot decided, 2: the last call's outcome, 0: the record form answered after all
        assert bits == (0 if (round_ == 2 and shape == "many_exceptions_next_time") else (1 if round_ == 0 else 2)), (shape, round_, bits)
        assert ("dense_expand" in ran) == (not (round_ == 2 and shape == "many_exceptions_next_time")), (shape, round_, ran)
    monkeypatch.setenv("MDB_JOIN_BITS", "0")
    k2, c2, f2, j2 = dev.join_group_count(dl, None, dr, None)
    assert j2 == ej and torch.equal(k2, k) and torch.equal(c2, c) and torch.equal(f2, f)


@pytest.mark.parametrize("shape", ["left_keys_beyond_the_right_tables_range", "null_left_keys", "null_right_keys"])
def test_one_pass_4096_bits_form_needs_every_left_row_at_the_leaf(dev, forced, shape):
    """The bit-per-left-row form looks at the bit of every left row that REACHES the leaf kernel: rows dropped on the way (NULL keys, keys
    outside the right table's window) would keep a set bit.  NULL-keyed left columns do not take the form; otherwise rows that did not
    arrive show as groups + cleared bits != left rows and the record form answers.  Results are the oracle's either way."""
    rng = np.random.default_rng(len(shape) + 3)
    n = 4_250_000 + 4096 * len(shape)
    kl = rng.permutation(1 << 25)[:n].astype(np.int64) + 50_000
    kr = rng.permutation(kl)
    nl = nr = None
    if shape == "left_keys_beyond_the_right_tables_range":
        kl[rng.integers(0, n, 60_000)] += 1 << 25        # (1.4 % of the left rows: the pilot still says "nearly every row a group")
    elif shape == "null_left_keys":
        nl = rng.random(n) < 0.01
    else:
        nr = rng.random(n) < 0.01
    _check(dev, kl, nl, kr, nr, expect_form=None, rounds=3)
