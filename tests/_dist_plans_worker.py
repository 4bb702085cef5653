"""Worker for tests/test_dist_gpu.py: every sharded entry point of include/mdb_dist.h at WORLD SIZE 4 AND 8 ON ONE GPU - N contexts
on one device, the bytes moved by the gloo test transport of tests/_dist_gpu_worker.py - on the plan shapes BASELINE configs[3] and
configs[4] land on at 8 GPUs (reference shapes: the join and GROUP BY loops of src/engine/executor_select.c:1076-1149, 1234-1280,
1526-1588 over tables whose rows are spread over the ranks).

mdb_shard_plan_make() branches on the world size (digits per rank = 512 / world: below 128 the receiver always runs a level of its
own; 4096-digit senders; three- and four-table leaves of half the values), and only world 1 (RCCL) and world 2 ever ran before.
Every case asserts WHICH branch ran (mdb_dist_last_plan: first-level digit bits, the receiver's own bits, leaf bits, bytes per word)
and checks every rank's groups against the numpy oracle restricted to the keys that hash to it.

  shapes : the operators called directly (join_group_count, _multi, group_count_keys, join_pairs keys-only and with payload,
           shuffle_rows), skew -> fallback on every rank, a key window of 2^30 values (configs[3]: 10^9 unique keys)
  sql    : query_execute() in sharded mode: the 17 statement shapes of the world-2 test, then configs[4]'s own statement - three
           tables with DOUBLE payload joined on one key + GROUP BY, and its join-only form - against numpy; VARCHAR columns whose
           cells cross the ranks as ids of a common dictionary
  fault  : one rank's first level fails (MDB_DIST_FAULT): every rank must return an error, none may hang; the call after works
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import np_oracle as orc  # noqa: E402
from midoridb_amd.dev import DeviceCtx  # noqa: E402
from midoridb_amd.dist import DistError, WIRE_32  # noqa: E402
from _dist_gpu_worker import gloo_transport, owned, check_shuffle_and_join, sharded_sql  # noqa: E402


def plan_of(dx):
    p = dx.last_plan()
    return None if p is None else (p["digit_bits"], p["receiver_bits"], p["leaf_bits"], p["word_bytes"])


def expect_plan(dx, world, want):
    p = dx.last_plan()
    got = plan_of(dx)
    assert p is not None and p["completed"] == 1 and p["world"] == world, p
    assert got == want[world], (world, got, want[world])
    assert p["digits_per_rank"] == (1 << p["digit_bits"]) // world
    # what one rank sends to one peer: the fixed-size blocks of every table + their region counters
    assert p["bytes_per_peer"] >= sum(p["block_bytes"]) > 0


def pool_keys(rng, base, span, total, dup):
    """keys drawn from a pool of values spread over [base, base + span): both tables draw from it, so most keys meet"""
    pool = base + rng.integers(0, span, max(total // max(dup, 1), 16), dtype=np.int64)
    return pool[rng.integers(0, len(pool), total)], pool[rng.integers(0, len(pool), total // dup + 7)]


def two_tables(dx, dev, world, rank, rng):
    # (rows per rank, key span, duplication, NULLs?, plan at world 4, plan at world 8) - plan = (digit bits, receiver bits, leaf bits, word bytes)
    cases = [
        (60_000, 40_000, 3, True, {4: (9, 0, 7, 2), 8: (9, 2, 5, 4)}),            # 2^16 values: one level at 128 digits per rank, the receiver's own level at 64
        (150_000, 3_000_000, 1, False, {4: (9, 0, 13, 2), 8: (9, 2, 11, 4)}),     # 2^22
        (200_000, 60_000_000, 2, True, {4: (12, 0, 14, 2), 8: (12, 0, 14, 2)}),   # 2^26: 4096 first-level digits, 64 segments per leaf at world 8
        (150_000, 130_000_000, 1, False, {4: (12, 0, 15, 2), 8: (12, 0, 15, 2)}),  # 2^27
        (150_000, 260_000_000, 1, False, {4: (9, 6, 13, 4), 8: (9, 6, 13, 4)}),   # 2^28: two levels, 4-byte words
        (200_000, (1 << 30) - 5, 2, True, {4: (9, 8, 13, 4), 8: (9, 8, 13, 4)}),  # 2^30: BASELINE configs[3]'s window (10^9 keys)
    ]
    for n, span, dup, with_nulls, want in cases:
        total = n * world
        base = -3_000_000
        ga, gb = pool_keys(rng, base, span, total, dup)
        na = (rng.random(len(ga)) < 0.01) if with_nulls else None
        nb = (rng.random(len(gb)) < 0.01) if with_nulls else None
        # uneven shards: rank 0 holds a third of its share of A; the last rank no row of B at all when there are NULLs
        cut = [0] + [int(len(ga) * (r + 1) / world * (0.33 if r == 0 else 1.0)) for r in range(world)]
        cut[-1] = len(ga)
        cutb = [0] + [int(len(gb) * (r + 1) / world) for r in range(world)]
        if with_nulls:
            cutb[-2] = cutb[-1]
        la, lb = slice(cut[rank], cut[rank + 1]), slice(cutb[rank], cutb[rank + 1])
        dx.set_key_ranges((base - 11, base + span + 11), (base, base + span - 1))
        out = (torch.empty(total + 8, dtype=torch.int64, device=dev.device), torch.empty(total + 8, dtype=torch.int64, device=dev.device))
        k, c, j = dx.join_group_count(dev.to_dev(ga[la]), dev.nullbits_dev(na[la]) if with_nulls else None, dev.to_dev(gb[lb]),
                                      dev.nullbits_dev(nb[lb]) if with_nulls else None, out=out)
        assert dx.last_fused(), (n, span)
        expect_plan(dx, world, want)
        ek, ec, _, ej = orc.join_group_count(ga, na, gb, nb)
        mine = owned(dx, ek, world, rank, gb, promised=(base, base + span - 1))
        got = dict(zip(k.cpu().numpy().tolist(), c.cpu().numpy().tolist()))
        assert len(got) == k.numel() and got == dict(zip(ek[mine].tolist(), ec[mine].tolist())), (n, span, len(got), int(mine.sum()))
        assert j == int(ec[mine].sum()) and dx.allreduce_sum([j])[0] == ej
    # the phases of the last call can be asked for (bench.py --gpus N prints them per rank)
    dx.set_phase_timing(True)
    k, c, j = dx.join_group_count(dev.to_dev(ga[la]), dev.nullbits_dev(na[la]), dev.to_dev(gb[lb]), dev.nullbits_dev(nb[lb]), out=out)
    ph = dx.last_phases()
    assert ph["device_ms"] > 0 and ph["first_level_ms"] > 0 and ph["receiver_ms"] >= 0, ph
    dx.set_phase_timing(False)


def keys_only_join(dx, dev, world, rank, rng):
    """BASELINE configs[3] (SELECT * over two key columns, unique keys, 10^9 of them over 8 GPUs): regions on the wire, every key
    written COUNT times; the window of 2^30 values is the plan 1.25 x 10^8 rows per GPU land on"""
    for n, span, want in ((150_000, (1 << 30) - 9, {4: (9, 8, 13, 4), 8: (9, 8, 13, 4)}),
                          (100_000, 900_000, {4: (9, 0, 11, 2), 8: (9, 2, 9, 4)})):
        total = n * world
        vals = np.unique(rng.integers(0, span, total + total // 8, dtype=np.int64))[:total] + 100
        ga = rng.permutation(vals)
        gb = rng.permutation(vals)[: len(vals) - 5]
        la = slice(len(ga) * rank // world, len(ga) * (rank + 1) // world)
        lb = slice(len(gb) * rank // world, len(gb) * (rank + 1) // world)
        dx.set_key_ranges((100, 100 + span - 1), (100, 100 + span - 1))
        key, _, _, J = dx.join_pairs(dev.to_dev(ga[la]), None, [], dev.to_dev(gb[lb]), None, [])
        assert dx.last_fused()
        expect_plan(dx, world, want)
        pl, pr = orc.join_pairs(ga, None, gb, None)
        mine = owned(dx, ga[pl], world, rank, gb, promised=(100, 100 + span - 1))
        assert J == int(mine.sum()) and np.array_equal(np.sort(key.cpu().numpy()), np.sort(ga[pl][mine]))
        assert dx.allreduce_sum([J])[0] == len(pl)
    # duplicates on both sides: every key COUNT times
    n = 40_000
    total = n * world
    ga = 7 + rng.integers(0, 30_000, total, dtype=np.int64)
    gb = 7 + rng.integers(0, 30_000, total, dtype=np.int64)
    la = slice(rank * n, (rank + 1) * n)
    dx.set_key_ranges((7, 30_006), (7, 30_006))
    key, _, _, J = dx.join_pairs(dev.to_dev(ga[la]), None, [], dev.to_dev(gb[la]), None, [])
    assert dx.last_fused()
    pl, pr = orc.join_pairs(ga, None, gb, None)
    mine = owned(dx, ga[pl], world, rank, gb, promised=(7, 30_006))
    assert J == int(mine.sum()) and np.array_equal(np.sort(key.cpu().numpy()), np.sort(ga[pl][mine]))


def several_tables(dx, dev, world, rank, rng):
    """three and four tables on one key in ONE exchange (BASELINE configs[4]): leaves of half the values per table"""
    cases = [
        (60_000, 50_000, 2, {4: (9, 0, 7, 2), 8: (9, 2, 5, 4)}),
        (150_000, 40_000_000, 2, {4: (9, 5, 12, 4), 8: (9, 5, 12, 4)}),           # 2^26 values, three tables: never the 4096-digit form
        (50_000, 20_000, 3, {4: (9, 0, 6, 2), 8: (9, 2, 4, 4)}),
        (100_000, 9_000_000, 3, {4: (9, 3, 12, 4), 8: (9, 3, 12, 4)}),            # 2^24 values, four tables
    ]
    for n, span, nright, want in cases:
        total = n * world
        base = -7_000
        pool = base + rng.integers(0, span, total // 2, dtype=np.int64)
        ga = pool[rng.integers(0, len(pool), total)]
        rights = [pool[rng.integers(0, len(pool), total // (t + 1) + 11)] for t in range(nright)]
        rights[-1][:100] = base - 5 - np.arange(100)		# keys of a further table outside the window: they join nothing, silently
        la = slice(rank * n, (rank + 1) * n)
        cuts = [slice(len(r) * rank // world, len(r) * (rank + 1) // world) for r in rights]
        dx.set_key_ranges((base, base + span - 1), (base, base + span - 1))
        got = dx.join_group_count_multi(dev.to_dev(ga[la]), None, [(dev.to_dev(r[c]), None) for r, c in zip(rights, cuts)])
        assert got is not None and dx.last_fused(), (n, span, nright)
        expect_plan(dx, world, want)
        k, c, j = got
        ek, ec, ef, _ = orc.join_group_count(ga, None, rights[0], None)
        for r in rights[1:]:
            k2, c2, f2, _ = orc.join_group_count(ek, None, r, None)
            ek, ec = k2, ec[f2] * c2
        mine = owned(dx, ek, world, rank, rights[0], promised=(base, base + span - 1))
        res = dict(zip(k.cpu().numpy().tolist(), c.cpu().numpy().tolist()))
        assert len(res) == k.numel() and res == dict(zip(ek[mine].tolist(), ec[mine].tolist())), (n, span, nright)
        assert dx.allreduce_sum([j])[0] == int(ec.sum())


def one_table(dx, dev, world, rank, rng):
    """GROUP BY of ONE sharded column as (key, COUNT) pairs: regions of the key column on the wire, nothing else"""
    for n, span, dup, want in ((80_000, 50_000, 1, {4: (9, 0, 7, 2), 8: (9, 2, 5, 4)}),
                               (150_000, 3_000_000, 2, {4: (9, 0, 13, 2), 8: (9, 2, 11, 4)}),
                               (150_000, 90_000_000, 1, {4: (12, 0, 15, 2), 8: (12, 0, 15, 2)}),
                               (100_000, 1_000_000_000, 2, {4: (9, 8, 13, 4), 8: (9, 8, 13, 4)})):
        total = n * world
        base = -123_456
        gk = base + rng.integers(0, span, total // dup, dtype=np.int64).repeat(dup)
        rng.shuffle(gk)
        mine_rows = slice(rank * len(gk) // world, (rank + 1) * len(gk) // world)
        dx.set_key_ranges((base, base + span - 1), (base, base + span - 1))
        got = dx.group_count_keys(dev.to_dev(gk[mine_rows]))
        assert got is not None and dx.last_fused(), (n, span)
        expect_plan(dx, world, want)
        vals, cnt = np.unique(gk, return_counts=True)
        own = orc.dest_of_fused(vals, world, base, span) == rank
        res = dict(zip(got[0].cpu().numpy().tolist(), got[1].cpu().numpy().tolist()))
        assert len(res) == got[0].numel() and res == dict(zip(vals[own].tolist(), cnt[own].tolist())), (n, span, len(res), int(own.sum()))


def skew_falls_back_everywhere(dx, dev, world, rank, rng):
    n = 100_000
    ga = 1000 + rng.integers(0, 50_000, n * world, dtype=np.int64)
    ga[rng.random(len(ga)) < 0.9] = 1234
    gb = 1000 + rng.integers(0, 50_000, n * world, dtype=np.int64)
    la = slice(rank * n, (rank + 1) * n)
    dx.set_key_ranges((1000, 51_000), (1000, 51_000))
    out = (torch.empty(n * world + 8, dtype=torch.int64, device=dev.device), torch.empty(n * world + 8, dtype=torch.int64, device=dev.device))
    k, c, j = dx.join_group_count(dev.to_dev(ga[la]), None, dev.to_dev(gb[la]), None, out=out)
    ek, ec, _, ej = orc.join_group_count(ga, None, gb, None)
    assert not dx.last_fused()
    mine = orc.dest_of(ek, world) == rank
    assert dict(zip(k.cpu().numpy().tolist(), c.cpu().numpy().tolist())) == dict(zip(ek[mine].tolist(), ec[mine].tolist()))
    assert dx.allreduce_sum([j])[0] == ej
    dx.set_key_ranges(None, None)


def shapes(world, rank):
    dev = DeviceCtx(0)
    dx = gloo_transport(dev, world, rank)
    dx.set_wire(WIRE_32)
    rng = np.random.default_rng(2024)		# (the same stream on every rank: every rank builds the same global tables)
    two_tables(dx, dev, world, rank, rng)
    keys_only_join(dx, dev, world, rank, rng)
    several_tables(dx, dev, world, rank, rng)
    one_table(dx, dev, world, rank, rng)
    skew_falls_back_everywhere(dx, dev, world, rank, rng)
    # rows with payload (INT64 with NULLs, DOUBLE) by destination, and the materialising join on what arrived
    check_shuffle_and_join(dx, dev, world, rank, 20_000, 31)
    dx.close()
    dev.close()


def fault(world, rank):
    """MDB_DIST_FAULT=first_level:<rank>: that rank's first partition level "fails"; it still posts blocks nobody reads and region
    counters that say so, every receiver raises the flag, the status exchange spreads it: an error on EVERY rank, within the test's
    timeout, and the handle keeps working afterwards"""
    dev = DeviceCtx(0)
    dx = gloo_transport(dev, world, rank)
    dx.set_wire(WIRE_32)
    rng = np.random.default_rng(5)
    n = 50_000
    ga = rng.integers(0, 200_000, n * world, dtype=np.int64)
    gb = rng.integers(0, 200_000, n * world, dtype=np.int64)
    la = slice(rank * n, (rank + 1) * n)
    dx.set_key_ranges((0, 199_999), (0, 199_999))
    ek, ec, _, ej = orc.join_group_count(ga, None, gb, None)
    out = (torch.empty(n * world + 8, dtype=torch.int64, device=dev.device), torch.empty(n * world + 8, dtype=torch.int64, device=dev.device))
    bad = world - 1
    for attempt in range(2):
        os.environ["MDB_DIST_FAULT"] = f"first_level:{bad}"
        try:
            dx.join_group_count(dev.to_dev(ga[la]), None, dev.to_dev(gb[la]), None, out=out)
            raise SystemExit(f"rank {rank}: the call must fail on every rank")
        except DistError as e:
            msg = str(e)
            assert ("fault injected" in msg) if rank == bad else ("a peer failed" in msg), msg
        del os.environ["MDB_DIST_FAULT"]
        k, c, j = dx.join_group_count(dev.to_dev(ga[la]), None, dev.to_dev(gb[la]), None, out=out)
        assert dx.last_fused()
        mine = owned(dx, ek, world, rank, gb, promised=(0, 199_999))
        assert dict(zip(k.cpu().numpy().tolist(), c.cpu().numpy().tolist())) == dict(zip(ek[mine].tolist(), ec[mine].tolist()))
        bad = 0
    # ranks that are not in the same collective call find out from the count exchange, before anything is posted
    try:
        if rank == 0:
            dx.group_count_keys(dev.to_dev(ga[la]))		# (a different kind of call than the peers make)
        else:
            dx.join_group_count(dev.to_dev(ga[la]), None, dev.to_dev(gb[la]), None, out=out)
        raise SystemExit("a call paired with another kind of call must fail")
    except DistError as e:
        assert "not in the same collective call" in str(e), str(e)
    k, c, j = dx.join_group_count(dev.to_dev(ga[la]), None, dev.to_dev(gb[la]), None, out=out)	# ... and nothing was left half-posted
    assert dx.last_fused() and dx.allreduce_sum([j])[0] == ej
    dx.close()
    dev.close()


def config5_sql(world, rank):
    """BASELINE configs[4]: A(id_a, x DOUBLE) JOIN B(id_b, y DOUBLE) JOIN C(id_c, z INT) on one key + GROUP BY id_a COUNT(*) through
    query_execute() in sharded mode (the three tables in ONE exchange), and its join-only form with x, y, z carried as payload"""
    from midoridb_amd.query import DB
    from midoridb_amd.dist import DatabaseDevice
    rng = np.random.default_rng(11)
    n = 30_000 * world
    dom = n // 3
    ka, kb, kc = rng.integers(0, dom, n), rng.integers(0, dom, n // 2), rng.integers(0, dom, n // 4)
    x, y = rng.standard_normal(n), rng.standard_normal(len(kb))
    x[::53] = -0.0
    z = rng.integers(-5, 5, len(kc))
    with DB() as db:
        dx = gloo_transport(DatabaseDevice(db, 0), world, rank)
        dx.attach_to_database(db)
        for ddl in ("CREATE TABLE A (id_a INT, x DOUBLE);", "CREATE TABLE B (id_b INT, y DOUBLE);", "CREATE TABLE C (id_c INT, z INT);"):
            db.execute(ddl)
        for name, cols in (("A", (ka, x)), ("B", (kb, y)), ("C", (kc, z))):
            m = len(cols[0])
            lo, hi = m * rank // world, m * (rank + 1) // world
            db.append_columns(name, [c[lo:hi] for c in cols])
        r = db.query("SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c GROUP BY id_a;")
        ek, ec, _, _ = orc.join_group_count(ka, None, kb, None)
        k2, c2, f2, _ = orc.join_group_count(ek, None, kc, None)
        exp = dict(zip(k2.tolist(), (ec[f2] * c2).tolist()))
        parts = [None] * world
        dist.all_gather_object(parts, list(zip(r.columns[r.names.index("A.id_a")].tolist(), r.columns[r.names.index("COUNT(*)")].tolist())))
        got = [kv for p in parts for kv in p]
        assert len(got) == len(exp) and dict(got) == exp, (len(got), len(exp))
        # the join-only form: every joined row with its DOUBLE bits, as a multiset over the ranks
        r = db.query("SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c;")
        pl, pr = orc.join_pairs(ka, None, kb, None)
        p2, pc = orc.join_pairs(ka[pl], None, kc, None)
        il, ir = pl[p2], pr[p2]
        exp_rows = sorted(zip(ka[il].tolist(), x[il].view(np.int64).tolist(), y[ir].view(np.int64).tolist(), z[pc].tolist()))
        cols = {nm: r.columns[i] for i, nm in enumerate(r.names)}
        mine = list(zip(cols["A.id_a"].tolist(), cols["A.x"].tolist(), cols["B.y"].tolist(), cols["C.z"].tolist()))	# (8-byte cells: a DOUBLE's bits)
        assert cols["B.id_b"].tolist() == cols["A.id_a"].tolist() == cols["C.id_c"].tolist()
        parts = [None] * world
        dist.all_gather_object(parts, mine)
        got_rows = sorted(t for p in parts for t in p)
        assert len(got_rows) == len(exp_rows), (len(got_rows), len(exp_rows))
        bad = [(g, e) for g, e in zip(got_rows, exp_rows) if g != e]
        assert not bad, (len(bad), bad[:3])
        assert db.query("SELECT COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c;").rows() == [(len(exp_rows),)]
    dist.barrier()


def varchar_sql(world, rank):
    """VARCHAR columns in sharded mode: a cell is the id of its string in ONE process's dictionary, so before cells cross the ranks the
    dictionaries are made known to each other and the cells travel as ids of a common dictionary every rank builds alike (shard_dict_sync).
    Ranks load different rows - and so hold different dictionaries with different ids for the same strings: joins on a VARCHAR key, GROUP BY
    and DISTINCT over one, VARCHAR payload behind an INT join, a string literal in WHERE - against plain Python over the whole tables"""
    import collections
    from midoridb_amd.query import DB
    from midoridb_amd.dist import DatabaseDevice
    rng = np.random.default_rng(23)
    names = [f"name-{i:04d}" for i in range(400)] + ["", "x", "a rather long string, 31 chars.."]
    npq, nq = 3000 * world, 1100 * world
    pn = [names[i] for i in rng.integers(0, len(names), npq)]
    pv = rng.integers(0, 300, npq)
    qn = [names[i] for i in rng.integers(50, len(names), nq)]		# (Q never holds the first 50 names, P never ... some of the last)
    qw = rng.integers(0, 300, nq)
    with DB() as db:
        dx = gloo_transport(DatabaseDevice(db, 0), world, rank)
        dx.attach_to_database(db)
        db.execute("CREATE TABLE P (name VARCHAR(40), v INT);")
        db.execute("CREATE TABLE Q (name2 VARCHAR(40), w INT);")
        # uneven, ORDER-SHUFFLED shards: every rank interns the strings in another order, so the same string has different ids everywhere
        lo, hi = npq * rank // world, npq * (rank + 1) // world
        order = rng.permutation(hi - lo) if rank % 2 else np.arange(hi - lo)[::-1]
        db.append_columns("P", [[pn[lo + i] for i in order], pv[lo:hi][order]])
        lo2, hi2 = nq * rank // world, nq * (rank + 1) // world
        db.append_columns("Q", [qn[lo2:hi2], qw[lo2:hi2]])

        def gathered(sql):
            r = db.query(sql)
            parts = [None] * world
            dist.all_gather_object(parts, [tuple(x.item() if hasattr(x, "item") else x for x in row) for row in zip(*[c.tolist() for c in r.columns])])
            return r.names, sorted(t for p in parts for t in p)
        nm, got = gathered("SELECT name, COUNT(*) FROM P GROUP BY name;")
        cnt = collections.Counter(pn)
        assert got == sorted((cnt[k], k) if nm[0] == "COUNT(*)" else (k, cnt[k]) for k in cnt), (len(got), len(cnt))
        nm, got = gathered("SELECT DISTINCT name FROM P;")
        assert got == sorted((k,) for k in cnt)
        nm, got = gathered("SELECT name, COUNT(*) FROM P INNER JOIN Q ON P.name = Q.name2 GROUP BY name;")
        cq = collections.Counter(qn)
        exp = {k: cnt[k] * cq[k] for k in cnt if k in cq}
        assert got == sorted((exp[k], k) if nm[0] == "COUNT(*)" else (k, exp[k]) for k in exp), (len(got), len(exp))
        nm, got = gathered("SELECT name, v, w FROM P INNER JOIN Q ON P.name = Q.name2 WHERE v < 20;")
        byname = collections.defaultdict(list)
        for n2, w in zip(qn, qw.tolist()):
            byname[n2].append(w)
        cols = {c: i for i, c in enumerate(nm)}
        exp_rows = sorted(tuple({"P.name": n1, "P.v": v, "Q.w": w}[c] for c in nm) for n1, v in zip(pn, pv.tolist()) if v < 20 for w in byname.get(n1, ()))
        assert got == exp_rows, (len(got), len(exp_rows), cols)
        nm, got = gathered("SELECT name, name2 FROM P INNER JOIN Q ON P.v = Q.w WHERE name = 'name-0007';")	# VARCHAR payload on both sides, INT key
        byw = collections.defaultdict(list)
        for n2, w in zip(qn, qw.tolist()):
            byw[w].append(n2)
        exp_rows = sorted(tuple({"P.name": n1, "Q.name2": n2}[c] for c in nm) for n1, v in zip(pn, pv.tolist()) if n1 == "name-0007" for n2 in byw.get(v, ()))
        assert got == exp_rows, (len(got), len(exp_rows))
        # rows inserted later bring new strings: announced by the next statement that moves VARCHAR cells
        db.execute(f"INSERT INTO P VALUES ('late-{rank}', 1), ('late-all', 2);")
        nm, got = gathered("SELECT name, COUNT(*) FROM P GROUP BY name;")
        late = dict((t[nm.index("P.name")], t[nm.index("COUNT(*)")]) for t in got if str(t[nm.index("P.name")]).startswith("late-"))
        assert late == dict([(f"late-{r}", 1) for r in range(world)] + [("late-all", world)]), late
    dist.barrier()
    varchar_empty_rank(world, rank)


def varchar_empty_rank(world, rank):
    """A rank that holds NO row of a table with a VARCHAR column (no device column at all: mdb_table_sync_device skips empty tables) takes
    part in the dictionary exchange all the same - whether a statement moves VARCHAR cells follows from its text and the schema alone - and
    translates the ids that arrive (round 5; the advisor's finding on mdb_exec_shard.c): VARCHAR as the join key and as payload behind an INT
    key, the empty rank on either side"""
    import collections
    from midoridb_amd.query import DB
    from midoridb_amd.dist import DatabaseDevice
    names = [f"s{i:03d}" for i in range(90)]
    for empty_tab in ("P", "Q"):
        for empty_rank in sorted({0, world - 1}):
            rng = np.random.default_rng(100 + empty_rank)
            rows = {"P": [(names[i], int(v)) for i, v in zip(rng.integers(0, 90, 700 * world), rng.integers(0, 60, 700 * world))],
                    "Q": [(names[i], int(v)) for i, v in zip(rng.integers(20, 90, 300 * world), rng.integers(0, 60, 300 * world))]}
            with DB() as db:
                dx = gloo_transport(DatabaseDevice(db, 0), world, rank)
                dx.attach_to_database(db)
                db.execute("CREATE TABLE P (name VARCHAR(40), v INT);")
                db.execute("CREATE TABLE Q (name2 VARCHAR(40), w INT);")
                for tab in ("P", "Q"):
                    # the rows of `empty_tab` are dealt to every rank but `empty_rank`; the other table to all ranks
                    holders = [r for r in range(world) if not (tab == empty_tab and r == empty_rank)]
                    if rank in holders:
                        mine = rows[tab][holders.index(rank)::len(holders)]
                        db.append_columns(tab, [[t[0] for t in mine], np.array([t[1] for t in mine], dtype=np.int64)])

                def gathered(sql):
                    r = db.query(sql)
                    parts = [None] * world
                    dist.all_gather_object(parts, [tuple(x.item() if hasattr(x, "item") else x for x in row) for row in zip(*[c.tolist() for c in r.columns])])
                    return r.names, sorted(t for p in parts for t in p)
                cp, cq = collections.Counter(t[0] for t in rows["P"]), collections.Counter(t[0] for t in rows["Q"])
                nm, got = gathered("SELECT name, COUNT(*) FROM P INNER JOIN Q ON P.name = Q.name2 GROUP BY name;")	# VARCHAR key, fused plan
                exp = {k: cp[k] * cq[k] for k in cp if k in cq}
                assert got == sorted((exp[k], k) if nm[0] == "COUNT(*)" else (k, exp[k]) for k in exp), (empty_tab, empty_rank, len(got), len(exp))
                nm, got = gathered("SELECT name, v, w FROM P INNER JOIN Q ON P.name = Q.name2 WHERE v < 9;")		# VARCHAR key, rows materialised
                byname = collections.defaultdict(list)
                for n2, w in rows["Q"]:
                    byname[n2].append(w)
                exp_rows = sorted(tuple({"P.name": n1, "P.v": v, "Q.w": w}[c] for c in nm) for n1, v in rows["P"] if v < 9 for w in byname.get(n1, ()))
                assert got == exp_rows, (empty_tab, empty_rank, len(got), len(exp_rows))
                nm, got = gathered("SELECT name, name2 FROM P INNER JOIN Q ON P.v = Q.w WHERE v < 5;")			# VARCHAR payload, INT key
                byw = collections.defaultdict(list)
                for n2, w in rows["Q"]:
                    byw[w].append(n2)
                exp_rows = sorted(tuple({"P.name": n1, "Q.name2": n2}[c] for c in nm) for n1, v in rows["P"] if v < 5 for n2 in byw.get(v, ()))
                assert got == exp_rows, (empty_tab, empty_rank, len(got), len(exp_rows))
                nm, got = gathered("SELECT DISTINCT name2 FROM Q;")
                assert got == sorted((k,) for k in cq), (empty_tab, empty_rank)
            dist.barrier()


def main():
    mode = sys.argv[1]
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if mode == "shapes":
        shapes(world, rank)
    elif mode == "fault":
        fault(world, rank)
    elif mode == "sql":
        sharded_sql(world, rank)
        config5_sql(world, rank)
        varchar_sql(world, rank)
    else:
        raise SystemExit(f"unknown mode {mode}")
    dist.barrier()
    if rank == 0:
        print(f"plans {mode} world {world} ok", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
