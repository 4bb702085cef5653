"""Worker for tests/test_distributed_gloo.py: one rank of the hash-partition + all-to-all + local-join
pipeline (midoridb_amd/shuffle.py) on CPU tensors over gloo, with the oracle's partition / join functions
plugged in for the device operators.  Exits non-zero on any mismatch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import np_oracle as orc  # noqa: E402
from midoridb_amd.shuffle import DistributedJoinGroupCount, TableShuffle  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 20_000
    total = n * world
    a = orc.gen_keys(n, rank * n, total, 42, 0)
    b = orc.gen_keys(n, rank * n, total, 43, total // 16)

    def partition_fn(keys, out):
        k, counts = orc.partition_by_dest(keys.numpy(), None, world)
        out[:len(k)] = torch.from_numpy(k)
        return out[:len(k)], [int(c) for c in counts]

    def join_fn(ka, kb, out):
        k, c, f, j = orc.join_group_count(ka.numpy(), None, kb.numpy(), None)
        return torch.from_numpy(k), torch.from_numpy(c), torch.from_numpy(f), j

    def partition32_fn(keys, out):      # 4-byte wire format
        k, counts = orc.partition_by_dest(keys.numpy(), None, world)
        out[:len(k)] = torch.from_numpy(k.astype(np.int32))
        return out[:len(k)], [int(c) for c in counts]

    def widen_fn(src32, out):
        out[:src32.numel()] = src32.to(torch.int64)
        return out[:src32.numel()]

    # one piece (the default), several pieces (3 also exercises an uneven last piece), 8- and 4-byte keys
    for chunks, wire32 in ((1, False), (3, False), (None, False), (2, True)):
        pipe = DistributedJoinGroupCount(None, world, rank, n, partition_fn=partition32_fn if wire32 else partition_fn, join_fn=join_fn,
                                         device=torch.device("cpu"), chunks=chunks, wire32=wire32, widen_fn=widen_fn)
        g, j = pipe.run(torch.from_numpy(a), torch.from_numpy(b), None)
        if chunks == 1:
            k1, c1 = pipe.last[0].numpy().copy(), pipe.last[1].numpy().copy()
        else:       # same groups whatever the chunking (order inside a rank may differ)
            o1, o2 = np.argsort(k1, kind="stable"), np.argsort(pipe.last[0].numpy(), kind="stable")
            assert np.array_equal(k1[o1], pipe.last[0].numpy()[o2]) and np.array_equal(c1[o1], pipe.last[1].numpy()[o2])
    k, c, _ = pipe.last
    # every key this rank owns must hash to this rank; gather all results on rank 0 and compare with one big join
    assert np.all(orc.dest_of(k.numpy(), world) == rank)
    gathered = [None] * world
    dist.all_gather_object(gathered, (k.numpy(), c.numpy(), j))
    if rank == 0:
        ek, ec, ef, ej = orc.join_group_count(orc.gen_keys(total, 0, total, 42, 0), None, orc.gen_keys(total, 0, total, 43, total // 16), None)
        keys = np.concatenate([x[0] for x in gathered])
        cnts = np.concatenate([x[1] for x in gathered])
        assert sum(x[2] for x in gathered) == ej == total
        o1, o2 = np.argsort(keys, kind="stable"), np.argsort(ek, kind="stable")
        assert np.array_equal(keys[o1], ek[o2]) and np.array_equal(cnts[o1], ec[o2])
        assert len(np.unique(keys)) == len(keys)		# groups are disjoint across ranks
        print("gloo distributed join ok", len(keys), "groups", ej, "joined rows")
    payload_join(rank, world)
    dist.barrier()
    dist.destroy_process_group()


def payload_join(rank, world):
    """BASELINE config 5 shape on 2 ranks: A(id_a, x DOUBLE) JOIN B(id_b, y DOUBLE) JOIN C(id_c, z INT) on one key,
    every table shuffled with its payload and its origin id, joined locally (oracle operators), then checked on
    rank 0 against one big join of the unsharded tables."""
    n = 3_000
    total = n * world
    rng = np.random.default_rng(7)		# same stream on every rank: the global tables
    ga = rng.integers(0, total // 2, total)
    gb = rng.integers(0, total // 2, total)
    gc = rng.integers(0, total // 2, total)
    gx, gy, gz = rng.normal(0, 1, total), rng.normal(0, 1, total), rng.integers(0, 100, total)
    sl = slice(rank * n, (rank + 1) * n)

    def partition_fn(keys):
        k = keys.numpy()
        d = orc.dest_of(k, world)
        o = np.argsort(d, kind="stable")
        return torch.from_numpy(k[o]), [int(c) for c in np.bincount(d, minlength=world)], torch.from_numpy(o.astype(np.int32))

    def gather_fn(col, rows):
        return col[rows.to(torch.int64)]

    sh = TableShuffle(world, torch.device("cpu"), partition_fn, gather_fn)
    ka, (xa,), oa = sh.run(torch.from_numpy(ga[sl]), [torch.from_numpy(gx[sl])], with_origin=True, rank=rank)
    kb, (yb,), ob = sh.run(torch.from_numpy(gb[sl]), [torch.from_numpy(gy[sl])], with_origin=True, rank=rank)
    kc, (zc,), oc = sh.run(torch.from_numpy(gc[sl]), [torch.from_numpy(gz[sl])], with_origin=True, rank=rank)
    assert np.all(orc.dest_of(ka.numpy(), world) == rank) and np.all(orc.dest_of(kc.numpy(), world) == rank)
    # origin ids name the source rank and row: payload must be the source table's cell, bit for bit
    src = (oa.numpy() >> 32) * n + (oa.numpy() & 0xFFFFFFFF)
    assert np.array_equal(gx[src].view(np.int64), xa.numpy().view(np.int64)) and np.array_equal(ga[src], ka.numpy())
    l, r = orc.join_pairs(ka.numpy(), None, kb.numpy(), None)
    kab = ka.numpy()[l]
    p, q = orc.join_pairs(kab, None, kc.numpy(), None)
    rows = np.stack([kab[p], xa.numpy().view(np.int64)[l][p], yb.numpy().view(np.int64)[r][p], zc.numpy()[q]], axis=1)
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


if __name__ == "__main__":
    main()
