"""Worker for tests/test_distributed_gloo.py: one rank of the exchange protocol of mdb_dist.hip, modelled on CPU over gloo
(tests/_exchange_model.py, oracle operators in place of the HIP kernels).  Exits non-zero on any mismatch."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import np_oracle as orc  # noqa: E402
from tests import _exchange_model as xm  # noqa: E402


def north_star(rank, world):
    """BASELINE configs[2] shape, sharded: every rank's groups hash to it, the ranks' groups together are one big join's"""
    n = 20_000
    total = n * world
    a = orc.gen_keys(n, rank * n, total, 42, 0)
    b = orc.gen_keys(n, rank * n, total, 43, total // 16)
    k, c, j = xm.join_group_count(a, None, b, None)
    assert np.all(orc.dest_of(k, world) == rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, (k, c, j))
    if rank == 0:
        ek, ec, ef, ej = orc.join_group_count(orc.gen_keys(total, 0, total, 42, 0), None, orc.gen_keys(total, 0, total, 43, total // 16), None)
        keys = np.concatenate([x[0] for x in gathered])
        cnts = np.concatenate([x[1] for x in gathered])
        assert sum(x[2] for x in gathered) == ej == total
        o1, o2 = np.argsort(keys, kind="stable"), np.argsort(ek, kind="stable")
        assert np.array_equal(keys[o1], ek[o2]) and np.array_equal(cnts[o1], ec[o2])
        assert len(np.unique(keys)) == len(keys)		# groups are disjoint across ranks
        print("gloo distributed join ok", len(keys), "groups", ej, "joined rows")


def shuffle_with_payload(rank, world):
    """mdb_dist_shuffle_rows: keys with NULLs, an INT64 column with NULLs and a DOUBLE column, read through a row-id vector;
    NULL keys dropped, or kept together on one rank; a failure on one rank reaches every rank through the counts"""
    n = 5_000
    rng = np.random.default_rng(11)		# same stream on every rank: the global table
    total = n * world
    gk = rng.integers(-20, total // 3, total)
    gkn = rng.random(total) < 0.05
    gf, gfn = rng.integers(-2**40, 2**40, total), rng.random(total) < 0.1
    gx = rng.standard_normal(total)
    sl = slice(rank * n, (rank + 1) * n)
    rid = rng.permutation(n)[: n * 3 // 4] if rank == 0 else np.arange(n)		# rank 0's stream is a filtered, reordered one
    sk, skn = gk[sl][rid], gkn[sl][rid]
    cols = [(gk[sl], gkn[sl], rid), (gf[sl], gfn[sl], rid), (gx[sl], None, rid)]
    streams = [None] * world
    dist.all_gather_object(streams, rank * n + rid)
    gsel = np.concatenate(streams)
    for flags in (0, xm.KEEP_NULL_KEYS):
        out, got = xm.shuffle_rows(sk, skn, cols, flags)
        dest = orc.dest_of(gk[gsel], world)
        if flags:
            dest = np.where(gkn[gsel], orc.dest_of(np.zeros(1, dtype=np.int64), world)[0], dest)
            mine = dest == rank
        else:
            mine = (dest == rank) & ~gkn[gsel]
        want = gsel[mine]
        assert got == len(want)
        # received order = (source rank, send order): compare as multisets of (key, key-null, f, f-null, x bits)
        def rows(k, kn, f, fn, x):
            return sorted(zip((np.where(kn, 0, k)).tolist(), kn.tolist(), np.where(fn, 0, f).tolist(), fn.tolist(), x.view(np.int64).tolist()))
        kn_got = out[0][1] if out[0][1] is not None else np.zeros(got, dtype=bool)
        assert rows(out[0][0], kn_got, out[1][0], out[1][1], out[2][0]) == rows(gk[want], gkn[want] & bool(flags), gf[want], gfn[want], gx[want])
        assert out[2][1] is None			# no rank has NULL bits for x: no bitmap comes back
    try:
        xm.shuffle_rows(sk, skn, cols, 0, fail=(rank == world - 1))
        raise SystemExit("a failed rank must fail the exchange everywhere")
    except xm.ExchangeFailed as e:
        assert f"rank {world - 1} failed" in str(e)
    if rank == 0:
        print("gloo shuffle rows ok")


def payload_join(rank, world):
    """BASELINE configs[4] shape on 2 ranks: A(id_a, x DOUBLE) JOIN B(id_b, y DOUBLE) JOIN C(id_c, z INT) on one key, every
    table shuffled with its payload, joined locally, checked on rank 0 against one big join of the unsharded tables."""
    n = 3_000
    total = n * world
    rng = np.random.default_rng(7)
    ga, gb, gc = (rng.integers(0, total // 2, total) for _ in range(3))
    gx, gy, gz = rng.normal(0, 1, total), rng.normal(0, 1, total), rng.integers(0, 100, total)
    sl = slice(rank * n, (rank + 1) * n)
    kab, (xa,), (yb,) = xm.join_pairs(ga[sl], None, [(gx[sl], None, None)], gb[sl], None, [(gy[sl], None, None)])
    assert np.all(orc.dest_of(kab, world) == rank)
    # the joined stream is already where its key hashes to: only C travels
    co, _ = xm.shuffle_rows(gc[sl], None, [(gc[sl], None, None), (gz[sl], None, None)])
    p, q = orc.join_pairs(kab, None, co[0][0], None)
    rows = np.stack([kab[p], xa[0][p], yb[0][p], co[1][0][q]], axis=1)
    gfirst, gcnt = orc.group_count(kab[p], None)
    gk = kab[p][gfirst]
    gathered = [None] * world
    dist.all_gather_object(gathered, (rows, gk, gcnt))
    if rank == 0:
        L, R = orc.join_pairs(ga, None, gb, None)
        P, Q = orc.join_pairs(ga[L], None, gc, None)
        want = np.stack([ga[L][P], gx.view(np.int64)[L][P], gy.view(np.int64)[R][P], gz[Q]], axis=1)
        got = np.concatenate([g[0] for g in gathered])
        assert got.shape == want.shape
        assert np.array_equal(got[np.lexsort(got.T[::-1])], want[np.lexsort(want.T[::-1])])		# same multiset of joined rows
        ef, ec = orc.group_count(ga[L][P], None)
        ek = ga[L][P][ef]
        k2 = np.concatenate([g[1] for g in gathered])
        c2 = np.concatenate([g[2] for g in gathered])
        o1, o2 = np.argsort(k2, kind="stable"), np.argsort(ek, kind="stable")
        assert np.array_equal(k2[o1], ek[o2]) and np.array_equal(c2[o1], ec[o2])
        print("gloo distributed payload join ok", len(got), "joined rows", len(ek), "groups")


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    north_star(rank, world)
    shuffle_with_payload(rank, world)
    payload_join(rank, world)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
