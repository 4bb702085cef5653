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
    def counts(send, n):
        cin = torch.tensor(send, dtype=torch.int64)
        cout = torch.empty(world * n, dtype=torch.int64)
        dist.all_to_all_single(cout, cin)
        return cout.tolist()

    def alltoallv(d_send, sc, sd, d_recv, rc, rd, es, _stream):
        torch.cuda.synchronize()
        dev.sync()
        nsend, nrecv = sum(sc), sum(rc)
        assert sd == [sum(sc[:i]) for i in range(world)] and rd == [sum(rc[:i]) for i in range(world)]
        hs = np.zeros(max(nsend, 1) * es, dtype=np.uint8)
        if nsend:
            assert dev.lib.mdb_dev_d2h(dev.h, hs.ctypes.data, d_send, nsend * es) == 0
        hr = torch.empty(max(nrecv, 1) * es, dtype=torch.uint8)
        dist.all_to_all_single(hr[:nrecv * es], torch.from_numpy(hs)[:nsend * es], [c * es for c in rc], [c * es for c in sc])
        if nrecv:
            assert dev.lib.mdb_dev_h2d(dev.h, d_recv, hr.numpy().ctypes.data, nrecv * es) == 0

    def allreduce(vals):
        t = torch.tensor(vals, dtype=torch.int64)
        dist.all_reduce(t)
        return t.tolist()

    return DistCtx.with_transport(dev, world, rank, counts, alltoallv, allreduce)


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
    mine = orc.dest_of(ek, world) == rank
    got = dict(zip(k.cpu().numpy().tolist(), c.cpu().numpy().tolist()))
    exp = dict(zip(ek[mine].tolist(), ec[mine].tolist()))
    assert len(got) == k.numel(), "duplicate group keys"
    assert got == exp, (len(got), len(exp))
    assert j == int(ec[mine].sum())
    assert dx.allreduce_sum([j])[0] == ej
    if expect_wire32 is not None:
        assert dx.last_wire32() == expect_wire32


def main():
    mode = sys.argv[1]
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) if mode == "rccl" else 0
    torch.cuda.set_device(local)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    mine = orc.dest_of(ek, world) == rank
    assert dict(zip(k.cpu().numpy().tolist(), c.cpu().numpy().tolist())) == dict(zip(ek[mine].tolist(), ec[mine].tolist()))
    assert dx.allreduce_sum([j])[0] == ej
    received = dx.allreduce_sum([dx.last_received_left()])[0]
    assert received <= n * world // 20 + 1, received		# (of the n * world left rows)
    # the same with catalog statistics instead of measured ranges (WIRE_32 + promised ranges: supersets are fine), and a promise
    # that does not hold: reported by the partition kernel on the rank that owns the offending key, never a wrong result
    dx.set_wire(WIRE_32)
    dx.set_key_ranges((10**6, 10**6 + n * world), (10**6 + n * world // 3 - 5, 10**6 + n * world // 3 + n * world // 20 + 5))
    k, c, j = dx.join_group_count(dev.to_dev(ga[la]), None, dev.to_dev(gb[lb]), None)
    assert dx.last_pruned()
    assert dict(zip(k.cpu().numpy().tolist(), c.cpu().numpy().tolist())) == dict(zip(ek[mine].tolist(), ec[mine].tolist()))
    assert dx.allreduce_sum([dx.last_received_left()])[0] <= n * world // 20 + 11
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
