"""Worker for tests/test_dist_gpu.py: the multi-GPU exchange through the C-ABI (include/mdb_dist.h).

  rccl : one rank per visible GPU over RCCL (world size 1 on the test box: still partition by destination, RCCL
         all-to-all with itself, split local join), both wire formats + the automatic choice
  gloo : TWO ranks on one GPU; struct mdb_dist_transport carries counts and keys through host memory with gloo, so the
         C exchange logic (counts, displacements, receive layout, overlap events, local join) runs with world size 2 on
         a one-GPU box.  Test infrastructure: the product's transport is RCCL.

Every rank checks its groups against the numpy oracle restricted to the keys that hash to it."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import np_oracle as orc  # noqa: E402
from midoridb_amd.dev import DeviceCtx  # noqa: E402
from midoridb_amd.dist import DistCtx, WIRE_32, WIRE_64, WIRE_AUTO  # noqa: E402


def gloo_transport(dev, world, rank):
    """world ranks on ONE GPU: the product's host-memory transport over this process group (midoridb_amd.dist.DistCtx.over_host_group)"""
    assert dist.get_world_size() == world and dist.get_rank() == rank
    return DistCtx.over_host_group(dev)


def owned(dx, keys, world, rank, gb, nb=None, promised=None):
    """which of `keys` this rank owns after the last join_group_count: by the hash of the key-by-destination path, or - when first-level
    regions travelled (mdb_dist_last_fused) - by the top bits of the window hash of the right table's global range"""
    if not dx.last_fused():
        return orc.dest_of(keys, world) == rank
    if promised is None:
        v = gb if nb is None else gb[~nb]
        promised = (int(v.min()), int(v.max()))
    return orc.dest_of_fused(keys, world, promised[0], promised[1] - promised[0] + 1) == rank


def check(dx, dev, world, rank, n, seed, lo, span, null_frac, expect_wire32):
    rng = np.random.default_rng(seed)
    total = n * world
    ga = lo + rng.integers(0, span, total, dtype=np.int64)
    gb = lo + rng.integers(0, span, total + 3 * world, dtype=np.int64)
    na = rng.random(len(ga)) < null_frac
    nb = rng.random(len(gb)) < null_frac
    # rank r holds an uneven slice (the last rank gets the remainder) of each table
    cut = [0] + [int(len(ga) * (r + 1) / world * (0.9 if r + 1 < world else 1.0)) for r in range(world)]
    cutb = [0] + [int(len(gb) * (r + 1) / world) for r in range(world)]
    la, lb = slice(cut[rank], cut[rank + 1]), slice(cutb[rank], cutb[rank + 1])
    k, c, j = dx.join_group_count(dev.to_dev(ga[la]), dev.nullbits_dev(na[la]) if null_frac else None, dev.to_dev(gb[lb]),
                                  dev.nullbits_dev(nb[lb]) if null_frac else None)
    ek, ec, _, ej = orc.join_group_count(ga, na, gb, nb)
    mine = owned(dx, ek, world, rank, gb, nb if null_frac else None)
    got = dict(zip(k.cpu().numpy().tolist(), c.cpu().numpy().tolist()))
    exp = dict(zip(ek[mine].tolist(), ec[mine].tolist()))
    assert len(got) == k.numel(), "duplicate group keys"
    assert got == exp, (len(got), len(exp))
    assert j == int(ec[mine].sum())
    assert dx.allreduce_sum([j])[0] == ej
    if expect_wire32 is not None and not dx.last_fused():
        assert dx.last_wire32() == expect_wire32


def _rows(cols):
    """sorted list of row tuples from [(values ndarray, nulls bool ndarray or None)]: NULL cells compare as None, DOUBLE cells by bits"""
    n = len(cols[0][0])
    out = []
    for v, nb in cols:
        v = np.asarray(v)
        bits = v.view(np.int64) if v.dtype == np.float64 else v.astype(np.int64)
        out.append([None if (nb is not None and nb[i]) else int(bits[i]) for i in range(n)])
    return sorted(zip(*out), key=lambda t: tuple((x is None, x or 0) for x in t))


def _host(dev, pair, n):
    from midoridb_amd.dev import unpack_nullbits
    v, nb = pair
    return v.cpu().numpy()[:n], (unpack_nullbits(nb.cpu().numpy(), n) if nb is not None else None)


def check_shuffle_and_join(dx, dev, world, rank, n, seed):
    """mdb_dist_shuffle_rows / mdb_dist_join_pairs against numpy: every rank must end up with exactly the rows (keys AND payload
    cells, DOUBLE bits and NULL flags included) whose key hashes to it - the materialising joins of BASELINE configs[3] / [4]"""
    from midoridb_amd.dist import KEEP_NULL_KEYS
    rng = np.random.default_rng(seed)
    total = n * world
    ka = rng.integers(-50, total // 2, total, dtype=np.int64)
    kb = rng.integers(-50, total // 2, total + 5 * world, dtype=np.int64)
    kan, kbn = rng.random(len(ka)) < 0.02, rng.random(len(kb)) < 0.02
    fa = rng.integers(-2**40, 2**40, len(ka), dtype=np.int64)
    fan = rng.random(len(ka)) < 0.1
    xa = rng.standard_normal(len(ka))
    xa[::97] = -0.0
    fb = rng.integers(0, 1000, len(kb), dtype=np.int64)
    cut = [0] + [int(len(ka) * (r + 1) / world * (0.8 if r + 1 < world else 1.0)) for r in range(world)]
    cutb = [0] + [int(len(kb) * (r + 1) / world) for r in range(world)]
    la, lb = slice(cut[rank], cut[rank + 1]), slice(cutb[rank], cutb[rank + 1])
    d_ka, d_kan = dev.to_dev(ka[la]), dev.nullbits_dev(kan[la])
    d_fa, d_fan, d_xa = dev.to_dev(fa[la]), dev.nullbits_dev(fan[la]), dev.to_dev(xa[la])
    d_kb, d_kbn, d_fb = dev.to_dev(kb[lb]), dev.nullbits_dev(kbn[lb]), dev.to_dev(fb[lb])
    # --- shuffle of table A: key + INT64 payload with NULLs + DOUBLE payload
    out, got = dx.shuffle_rows(d_ka, d_kan, [d_ka, (d_fa, d_fan), d_xa])
    mine = (orc.dest_of(ka, world) == rank) & ~kan
    exp = _rows([(ka[mine], None), (fa[mine], fan[mine]), (xa[mine], None)])
    res = _rows([_host(dev, out[0], got), _host(dev, out[1], got), _host(dev, out[2], got)])
    assert got == int(mine.sum()) and res == exp, (got, int(mine.sum()))
    # --- the same stream read through a row-id vector (a filtered / joined tuple stream), NULL keys kept (GROUP BY shuffle)
    nloc = cut[rank + 1] - cut[rank]
    rid = rng.permutation(nloc)[: max(nloc * 2 // 3, 1)].astype(np.uint32)
    d_rid = dev.to_dev(rid)
    sk, skn = dev.gather64(d_ka, d_kan, d_rid, len(rid))
    out, got = dx.shuffle_rows(sk, skn, [(d_ka, d_kan, d_rid), (d_fa, d_fan, d_rid), (d_xa, None, d_rid)], KEEP_NULL_KEYS)
    # what every rank's stream holds, globally
    allrid = [None] * world
    dist.all_gather_object(allrid, (cut[rank], rid.astype(np.int64)))
    gsel = np.concatenate([off + r for off, r in allrid])
    gk, gkn = ka[gsel], kan[gsel]
    dest = np.where(gkn, orc.dest_of(np.zeros(1, dtype=np.int64), world)[0], orc.dest_of(gk, world))
    mine = dest == rank
    exp = _rows([(gk[mine], gkn[mine]), (fa[gsel][mine], fan[gsel][mine]), (xa[gsel][mine], None)])
    res = _rows([_host(dev, out[0], got), _host(dev, out[1], got), _host(dev, out[2], got)])
    assert got == int(mine.sum()) and res == exp
    # --- sharded materialising join with payload on both sides
    key, lcols, rcols, J = dx.join_pairs(d_ka, d_kan, [(d_fa, d_fan), d_xa], d_kb, d_kbn, [d_fb])
    pl, pr = orc.join_pairs(ka, kan, kb, kbn)
    mine = orc.dest_of(ka[pl], world) == rank
    pl, pr = pl[mine], pr[mine]
    exp = _rows([(ka[pl], None), (fa[pl], fan[pl]), (xa[pl], None), (fb[pr], None)])
    res = _rows([(key.cpu().numpy(), None), _host(dev, lcols[0], J), _host(dev, lcols[1], J), _host(dev, rcols[0], J)])
    assert J == len(pl) and res == exp, (J, len(pl))
    assert dx.allreduce_sum([J])[0] == len(mine)
    # keys only (BASELINE configs[3]: SELECT * over two key columns), and an empty side
    key, _, _, J = dx.join_pairs(d_ka, None, [], d_kb, None, [])
    pl, pr = orc.join_pairs(ka, None, kb, None)
    assert sorted(key.cpu().numpy().tolist()) == sorted(ka[pl][orc.dest_of(ka[pl], world) == rank].tolist())
    e = torch.empty(0, dtype=torch.int64, device=dev.device)
    key, lc, rc, J = dx.join_pairs(d_ka, d_kan, [d_xa], e, None, [e])
    assert J == 0 and key.numel() == 0 and lc[0][0].numel() == 0


def sharded_sql(world, rank):
    """query_execute() in sharded mode at world size 2 on one GPU (mdb_database_set_dist with the test transport): every rank loads
    ITS rows, runs the same statements, and the ranks' results together must be exactly the rows oracle/naive.py computes over
    the whole tables - joins with payload (INT64 with NULLs, DOUBLE), three-way joins on one key and on two, WHERE conjuncts
    pushed below the exchange, GROUP BY of a non-key column, DISTINCT, global COUNT(*), cross and non-equi joins (the new table
    replicated on every rank)."""
    from oracle.naive import Naive
    from oracle.ref import sql_to_rpn
    from midoridb_amd.query import DB, QueryError
    rng = np.random.default_rng(77)
    na, nb, nc = 600, 700, 200	# (oracle/naive.py joins with nested Python loops)
    dom = 250

    def col(n, lo, hi, nullp):
        return rng.integers(lo, hi, n), rng.random(n) < nullp
    xa = rng.integers(-4, 5, na) * 0.5
    xa[rng.random(na) < 0.1] = -0.0
    yb = rng.integers(-4, 5, nb) * 0.5
    a = [col(na, 0, dom, 0.03), col(na, -50, 50, 0.1), (xa, rng.random(na) < 0.05)]
    b = [col(nb, 0, dom, 0.03), col(nb, 0, 40, 0.0), (yb, np.zeros(nb, dtype=bool))]
    c = [col(nc, 0, 60, 0.0), col(nc, 0, 9, 0.1)]
    ddl = ["CREATE TABLE A (id_a INT, f1 INT, x DOUBLE);", "CREATE TABLE B (id_b INT, f2 INT, y DOUBLE);", "CREATE TABLE C (id_c INT, f3 INT);"]
    queries = [
        "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b;",
        "SELECT f1, x, f2 FROM A INNER JOIN B ON A.id_a = B.id_b WHERE f1 > -20 AND f2 < 30;",
        "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c;",
        "SELECT id_a, f2, f3 FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON B.f2 = C.id_c WHERE f3 <> 4;",
        "SELECT id_a, id_b FROM A INNER JOIN B ON A.id_a = B.id_b AND f1 < f2;",
        "SELECT f1, y FROM A INNER JOIN B ON A.x = B.y WHERE id_a < 40 AND id_b < 40;",
        "SELECT f2, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b WHERE f1 > -20 GROUP BY f2;",
        "SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a;",
        "SELECT id_c, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c GROUP BY id_c;",
        "SELECT f1, COUNT(*) FROM A GROUP BY f1;",
        "SELECT f1, f3 FROM A INNER JOIN C ON A.id_a = C.id_c GROUP BY f1, f3;",
        "SELECT DISTINCT f1 FROM A;",
        "SELECT DISTINCT f2, f3 FROM B INNER JOIN C ON B.f2 = C.id_c;",
        "SELECT f2, COUNT(*) FROM B GROUP BY f2 HAVING COUNT(*) > 17;",
        "SELECT id_a, f1 FROM A WHERE f1 IS NULL;",
        "SELECT id_a, x FROM A INNER JOIN B ON A.id_a = B.id_b WHERE x > 0.4 ORDER BY id_a;",
        "SELECT id_b, COUNT(*) FROM B INNER JOIN C ON B.f2 = C.id_c GROUP BY id_b;",
        # joins without an equi-join key: the new table is replicated on every rank (mdb_dist_broadcast_rows), each rank pairs its own rows with it
        "SELECT id_a, id_c FROM A, C WHERE f1 > 40;",
        "SELECT id_a, f3 FROM A INNER JOIN C ON A.id_a < C.id_c AND f3 = 2;",
        "SELECT f3, COUNT(*) FROM A, C WHERE f1 > 45 GROUP BY f3;",
        "SELECT f1, f3 FROM A, C WHERE id_a < 6;",			# NULL cells on both sides of a replicated join
    ]
    counts = [
        "SELECT COUNT(*) FROM A WHERE f1 > 0;",
        "SELECT COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b WHERE f2 <= 10 OR f1 IS NULL;",
        "SELECT COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b;",
        "SELECT COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON B.f2 = C.id_c;",
        "SELECT COUNT(*) FROM A, C;",
    ]
    tables = {}
    with DB() as db:
        from midoridb_amd.dist import DatabaseDevice
        dx = gloo_transport(DatabaseDevice(db, 0), world, rank)
        dx.attach_to_database(db)		# the database owns the handle now
        for sdl, name, data in zip(ddl, "ABC", (a, b, c)):
            db.execute(sdl)
            n = len(data[0][0])
            lo, hi = n * rank // world, n * (rank + 1) // world
            if name == "C" and rank == 0:
                lo, hi = 0, 0		# one rank holds no row of C at all
            elif name == "C" and rank == 1:
                lo = 0			# (... rank 1 holds them as well)
            db.append_columns(name, [d[0][lo:hi] for d in data], [d[1][lo:hi] for d in data])
            cols = sdl[sdl.index("(") + 1:sdl.rindex(")")].split(",")
            tables[name] = ([x.split()[0] for x in cols],
                            [[None if d[1][i] else (float(d[0][i]) if d[0].dtype == np.float64 else int(d[0][i])) for d in data] for i in range(n)])
        ex = Naive(tables)
        for q in queries:
            names, rows = ex.run(sql_to_rpn(q))
            res = db.query(q)
            assert res.names == names, q
            parts = [None] * world
            dist.all_gather_object(parts, res.rows())
            got = sorted(r for p in parts for r in p)
            assert got == sorted(rows), (q, len(got), len(rows))
        for q in counts:
            names, rows = ex.run(sql_to_rpn(q))
            assert db.query(q).rows() == rows, q		# the global count on every rank
        # any order allowed: GROUP BY of one table ships first-level regions of the key column instead of rows (no NULL key on any
        # rank: f2; f1 has NULLs and keeps the row exchange) - the ranks' groups together are the oracle's, as a set
        db.groups_any_order(True)
        # ... and after a join the decision "no NULL key" comes from the exchanged stream's bitmap, not from catalog counters (the
        # shadow tables that arrive over the wire have none): f1 carries NULLs through the join, f2 does not
        for q in ("SELECT f2, COUNT(*) FROM B GROUP BY f2;", "SELECT f1, COUNT(*) FROM A GROUP BY f1;",
                  "SELECT f2, COUNT(*) FROM B GROUP BY f2 HAVING COUNT(*) > 17;", "SELECT COUNT(*) FROM B GROUP BY f2;",
                  "SELECT f1, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY f1;",
                  "SELECT f2, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY f2;",
                  "SELECT f3, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON B.f2 = C.id_c GROUP BY f3;"):
            names, rows = ex.run(sql_to_rpn(q))
            res = db.query(q)
            assert res.names == names, q
            parts = [None] * world
            dist.all_gather_object(parts, res.rows())
            assert sorted(r for p in parts for r in p) == sorted(rows), q
        db.groups_any_order(False)
        for q, what in (("SELECT f1 FROM A INNER JOIN B ON A.id_a = B.id_b LIMIT 3, 4;", "LIMIT"),):
            try:
                db.query(q)
                raise SystemExit(f"{q} must be refused in sharded mode")
            except QueryError as e:
                assert what in str(e), str(e)
    dist.barrier()
    if rank == 0:
        print(f"sharded sql world {world} ok", flush=True)


def fused_shapes(dx, dev, world, rank):
    """the regions-on-the-wire operator through its forms: one level (2-byte and 4-byte words), the receiver's own second level,
    duplicates on both sides, NULL keys, an empty right table on one rank, a rank with far fewer rows; and skewed keys that
    overflow a region (every rank takes the exact path together)"""
    rng = np.random.default_rng(99)
    dx.set_wire(WIRE_32)
    for n, span, dup in ((300_000, 40_000, 3), (500_000, 3_000_000, 1), (400_000, 60_000_000, 2), (1_500_000, 1_200_000, 1),
                         (600_000, 130_000_000, 1), (500_000, 17_000_000, 4)):   # (windows of 2^24 .. 2^27 values: 4096 first-level digits)
        total = n * world
        base = 5_000_000
        ga = base + rng.integers(0, span, total, dtype=np.int64)
        gb = base + rng.integers(0, span, total // dup + 7, dtype=np.int64)
        na, nb = rng.random(len(ga)) < 0.01, rng.random(len(gb)) < 0.01
        cut = [0] + [int(len(ga) * (r + 1) / world * (0.5 if r + 1 < world else 1.0)) for r in range(world)]
        cutb = [0] + [int(len(gb) * (r + 1) / world) for r in range(world)]
        la, lb = slice(cut[rank], cut[rank + 1]), slice(cutb[rank], cutb[rank + 1])
        dx.set_key_ranges((base - 3, base + span + 3), (base, base + span - 1))
        out = (torch.empty(total + 8, dtype=torch.int64, device=dev.device), torch.empty(total + 8, dtype=torch.int64, device=dev.device))
        k, c, j = dx.join_group_count(dev.to_dev(ga[la]), dev.nullbits_dev(na[la]), dev.to_dev(gb[lb]), dev.nullbits_dev(nb[lb]), out=out)
        assert dx.last_fused(), (n, span)
        ek, ec, _, ej = orc.join_group_count(ga, na, gb, nb)
        mine = owned(dx, ek, world, rank, gb, promised=(base, base + span - 1))
        got = dict(zip(k.cpu().numpy().tolist(), c.cpu().numpy().tolist()))
        assert len(got) == k.numel() and got == dict(zip(ek[mine].tolist(), ec[mine].tolist())), (n, span, len(got), int(mine.sum()))
        assert dx.allreduce_sum([j])[0] == ej
    # the join of two key columns alone (BASELINE configs[3]): regions on the wire, every key written COUNT times
    for n, span in ((200_000, 30_000), (700_000, 700_000)):
        total = n * world
        ga = 100 + (rng.permutation(total).astype(np.int64) if span == n else rng.integers(0, span, total, dtype=np.int64))
        gb = 100 + (rng.permutation(total).astype(np.int64)[: total - 5] if span == n else rng.integers(0, span, total, dtype=np.int64))
        la, lb = slice(rank * n, (rank + 1) * n), slice(len(gb) * rank // world, len(gb) * (rank + 1) // world)
        dx.set_key_ranges((100, 100 + total), (100, 100 + total))
        key, _, _, J = dx.join_pairs(dev.to_dev(ga[la]), None, [], dev.to_dev(gb[lb]), None, [])
        assert dx.last_fused()
        pl, pr = orc.join_pairs(ga, None, gb, None)
        mine = owned(dx, ga[pl], world, rank, gb, promised=(100, 100 + total))
        assert J == int(mine.sum()) and np.array_equal(np.sort(key.cpu().numpy()), np.sort(ga[pl][mine]))
    # three and four tables on one key in ONE exchange (BASELINE configs[4] shape): counts multiplied where the regions meet
    for n, span, nright in ((300_000, 50_000, 2), (600_000, 40_000_000, 2), (250_000, 20_000, 3)):
        total = n * world
        base = -7_000
        ga = base + rng.integers(0, span, total, dtype=np.int64)
        rights = [base + rng.integers(0, span, total // (t + 1) + 11, dtype=np.int64) for t in range(nright)]
        rights[-1][:100] = base - 5 - np.arange(100)		# keys of a further table outside the window: they join nothing, silently
        la = slice(rank * n, (rank + 1) * n)
        cuts = [slice(len(r) * rank // world, len(r) * (rank + 1) // world) for r in rights]
        dx.set_key_ranges((base, base + span - 1), (base, base + span - 1))
        got = dx.join_group_count_multi(dev.to_dev(ga[la]), None, [(dev.to_dev(r[c]), None) for r, c in zip(rights, cuts)])
        assert got is not None and dx.last_fused(), (n, span, nright)
        k, c, j = got
        ek, ec, ef, _ = orc.join_group_count(ga, None, rights[0], None)
        for r in rights[1:]:
            k2, c2, f2, _ = orc.join_group_count(ek, None, r, None)
            ek, ec = k2, ec[f2] * c2
        mine = owned(dx, ek, world, rank, rights[0], promised=(base, base + span - 1))
        res = dict(zip(k.cpu().numpy().tolist(), c.cpu().numpy().tolist()))
        assert len(res) == k.numel() and res == dict(zip(ek[mine].tolist(), ec[mine].tolist())), (n, span, nright)
        assert dx.allreduce_sum([j])[0] == int(ec.sum())
    # GROUP BY of ONE sharded column as (key, COUNT) pairs: regions of the key column on the wire, nothing else
    for n, span, dup in ((400_000, 50_000, 1), (700_000, 3_000_000, 2), (500_000, 90_000_000, 1)):
        total = n * world
        base = -123_456
        gk = base + rng.integers(0, span, total // dup, dtype=np.int64).repeat(dup)
        rng.shuffle(gk)
        mine_rows = slice(rank * len(gk) // world, (rank + 1) * len(gk) // world)
        dx.set_key_ranges((base, base + span - 1), (base, base + span - 1))
        got = dx.group_count_keys(dev.to_dev(gk[mine_rows]))
        assert got is not None and dx.last_fused(), (n, span)
        vals, cnt = np.unique(gk, return_counts=True)
        own = orc.dest_of_fused(vals, world, base, span) == rank
        res = dict(zip(got[0].cpu().numpy().tolist(), got[1].cpu().numpy().tolist()))
        assert len(res) == got[0].numel() and res == dict(zip(vals[own].tolist(), cnt[own].tolist())), (n, span, len(res), int(own.sum()))
    dx.set_key_ranges(None, None)
    assert dx.group_count_keys(dev.to_dev(gk[mine_rows])) is None        # ranges unknown: not served, on every rank
    assert dx.join_group_count_multi(dev.to_dev(ga[la]), None, [(dev.to_dev(r[c]), None) for r, c in zip(rights, cuts)]) is None	# ranges unknown: not served
    # 90 % of the left rows share one key: its region overflows on the sender -> all ranks fall back, same groups
    n = 400_000
    ga = 1000 + rng.integers(0, 50_000, n * world, dtype=np.int64)
    ga[rng.random(len(ga)) < 0.9] = 1234
    gb = 1000 + rng.integers(0, 50_000, n * world, dtype=np.int64)
    la = slice(rank * n, (rank + 1) * n)
    dx.set_key_ranges((1000, 51_000), (1000, 51_000))
    out = (torch.empty(n * world + 8, dtype=torch.int64, device=dev.device), torch.empty(n * world + 8, dtype=torch.int64, device=dev.device))
    k, c, j = dx.join_group_count(dev.to_dev(ga[la]), None, dev.to_dev(gb[la]), None, out=out)
    ek, ec, _, ej = orc.join_group_count(ga, None, gb, None)
    assert not dx.last_fused()
    assert dx.allreduce_sum([j])[0] == ej and dx.allreduce_sum([k.numel()])[0] == len(ek)
    dx.set_key_ranges(None, None)


def main():
    mode = sys.argv[1]
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) if mode == "rccl" else 0
    torch.cuda.set_device(local)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if mode == "sql":
        sharded_sql(world, rank)
        dist.destroy_process_group()
        return
    dev = DeviceCtx(local)
    dx = DistCtx.from_torch(dev) if mode == "rccl" else gloo_transport(dev, world, rank)
    # keys of 2 x 10^5 rows per rank: single-level local partition; 1.2 x 10^6: two levels, sampled narrow / compact forms
    for n, seed in ((200_000, 1), (1_200_000, 2)):
        dx.set_wire(WIRE_AUTO)
        check(dx, dev, world, rank, n, seed, 0, n * world // 2, 0.0, True)
        check(dx, dev, world, rank, n, seed + 10, -5, n * world // 3, 0.03, True)
        check(dx, dev, world, rank, n, seed + 20, 10**12, n * world, 0.0, False)	# keys beyond 32 bits: 8-byte wire format
        dx.set_wire(WIRE_64)
        check(dx, dev, world, rank, n, seed + 30, 0, n * world // 2, 0.01, False)
        dx.set_wire(WIRE_32)
        check(dx, dev, world, rank, n, seed + 40, -1000, n * world // 2, 0.0, True)
    for n, seed in ((3_000, 5), (400_000, 6)):
        check_shuffle_and_join(dx, dev, world, rank, n, seed)
    # min-max pruning before the shuffle (MDB_WIRE_AUTO): the right table's keys cover a twentieth of the left table's range -
    # left rows outside the right table's GLOBAL range stay home; the groups are the oracle's all the same
    dx.set_wire(WIRE_AUTO)
    rng = np.random.default_rng(4711)
    n = 600_000
    ga = rng.permutation(n * world).astype(np.int64) + 10**6
    gb = rng.integers(n * world // 3, n * world // 3 + n * world // 20, n * world, dtype=np.int64) + 10**6
    la, lb = slice(rank * n, (rank + 1) * n), slice(rank * n, (rank + 1) * n)
    k, c, j = dx.join_group_count(dev.to_dev(ga[la]), None, dev.to_dev(gb[lb]), None)
    ek, ec, _, ej = orc.join_group_count(ga, None, gb, None)
    mine = owned(dx, ek, world, rank, gb)
    assert dict(zip(k.cpu().numpy().tolist(), c.cpu().numpy().tolist())) == dict(zip(ek[mine].tolist(), ec[mine].tolist()))
    assert dx.allreduce_sum([j])[0] == ej
    mine = orc.dest_of(ek, world) == rank
    received = dx.allreduce_sum([dx.last_received_left()])[0]
    assert received <= n * world // 20 + 1, received		# (of the n * world left rows)
    # the same with catalog statistics instead of measured ranges (WIRE_32 + promised ranges: supersets are fine), and a promise
    # that does not hold: reported by the partition kernel on the rank that owns the offending key, never a wrong result
    dx.set_wire(WIRE_32)
    dx.set_key_ranges((10**6, 10**6 + n * world), (10**6 + n * world // 3 - 5, 10**6 + n * world // 3 + n * world // 20 + 5))
    k, c, j = dx.join_group_count(dev.to_dev(ga[la]), None, dev.to_dev(gb[lb]), None)
    assert dx.last_pruned()
    assert dx.last_fused()		# known ranges: the first partition level IS the exchange (mdb_dev_shard.hip, regions on the wire)
    mine_f = owned(dx, ek, world, rank, gb, promised=(10**6 + n * world // 3 - 5, 10**6 + n * world // 3 + n * world // 20 + 5))
    assert dict(zip(k.cpu().numpy().tolist(), c.cpu().numpy().tolist())) == dict(zip(ek[mine_f].tolist(), ec[mine_f].tolist()))
    assert dx.allreduce_sum([dx.last_received_left()])[0] <= n * world // 20 + 11
    os.environ["MDB_DIST_FUSED"] = "0"	# ... and the key-by-destination path on the same input
    k, c, j = dx.join_group_count(dev.to_dev(ga[la]), None, dev.to_dev(gb[lb]), None)
    assert not dx.last_fused()
    assert dict(zip(k.cpu().numpy().tolist(), c.cpu().numpy().tolist())) == dict(zip(ek[mine].tolist(), ec[mine].tolist()))
    del os.environ["MDB_DIST_FUSED"]
    fused_shapes(dx, dev, world, rank)
    dx.set_key_ranges((10**6, 10**6 + n * world), (10**6 + n * world // 3, 10**6 + n * world // 3 + 10))	# too tight for the right table
    try:
        dx.join_group_count(dev.to_dev(ga[la]), None, dev.to_dev(gb[lb]), None)
        raised = False
    except Exception as ex:
        raised = "promised" in str(ex)
    assert raised
    dx.set_key_ranges(None, None)
    # an empty table on one side, and a promise that does not hold
    dx.set_wire(WIRE_AUTO)
    e = torch.empty(0, dtype=torch.int64, device=dev.device)
    k, c, j = dx.join_group_count(dev.to_dev(np.arange(10, dtype=np.int64)), None, e, None)
    assert k.numel() == 0 and j == 0
    dx.set_wire(WIRE_32)
    try:
        dx.join_group_count(dev.to_dev(np.array([1, 2**40], dtype=np.int64)), None, dev.to_dev(np.array([1], dtype=np.int64)), None)
        raised = False
    except Exception as ex:
        raised = "4-byte wire format" in str(ex)
    # every rank must fail the same way (the bad key is on every rank's slice here)
    assert raised
    dx.close()
    dev.close()
    dist.barrier()
    if rank == 0:
        print(f"dist {mode} world {world} ok", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
