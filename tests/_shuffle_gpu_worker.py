"""Worker for tests/test_dev_ops_gpu.py::test_table_shuffle_payload_join_rccl: the multi-GPU exchange path of
BASELINE configs 4 / 5 (hash partition by destination WITH row ids -> payload gather -> RCCL all-to-all per column ->
local joins -> GROUP BY) with the device operators, on however many GPUs the launcher gives it (1 on the test box:
world size 1 still goes through mdb_dev_partition_by_dest, the gathers and RCCL).  Checked against the numpy oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import np_oracle as orc  # noqa: E402
from midoridb_amd.dev import DeviceCtx  # noqa: E402
from midoridb_amd.shuffle import DistributedJoinGroupCount, TableShuffle  # noqa: E402


def main():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    dev = DeviceCtx(local)
    n = 400_000
    total = n * world
    rng = np.random.default_rng(11)
    ga, gb, gc = (rng.integers(0, total // 2, total) for _ in range(3))
    gx, gy, gz = rng.normal(0, 1, total), rng.normal(0, 1, total), rng.integers(0, 100, total)
    sl = slice(rank * n, (rank + 1) * n)

    def partition_fn(keys):
        out, counts, rid = dev.partition_by_dest(keys, None, world, with_rid=True)
        return out, counts, rid

    def gather_fn(col, rows):
        return dev.gather64(col, None, rows, rows.numel())[0]

    sh = TableShuffle(world, dev.device, partition_fn, gather_fn)
    ka, (xa,), oa = sh.run(dev.to_dev(ga[sl]), [dev.to_dev(gx[sl])], with_origin=True, rank=rank)
    kb, (yb,), _ = sh.run(dev.to_dev(gb[sl]), [dev.to_dev(gy[sl])])
    kc, (zc,), _ = sh.run(dev.to_dev(gc[sl]), [dev.to_dev(gz[sl])])
    src = (oa.cpu().numpy() >> 32) * n + (oa.cpu().numpy() & 0xFFFFFFFF)
    assert np.array_equal(ga[src], ka.cpu().numpy()) and np.array_equal(gx[src].view(np.int64), xa.cpu().numpy().view(np.int64))
    assert np.all(orc.dest_of(ka.cpu().numpy(), world) == rank)
    # local (A join B) join C with payload, then GROUP BY key COUNT(*)
    l, r = dev.join_pairs(ka, None, kb, None)
    kab = dev.gather64(ka, None, l, l.numel())[0]
    p, q = dev.join_pairs(kab, None, kc, None)
    key = dev.gather64(kab, None, p, p.numel())[0]
    x3 = dev.gather64(xa, None, dev.gather32(l, p), p.numel())[0]
    y3 = dev.gather64(yb, None, dev.gather32(r, p), p.numel())[0]
    z3 = dev.gather64(zc, None, q, q.numel())[0]
    first, cnt = dev.group_count(key, None)
    rows = np.stack([key.cpu().numpy(), x3.cpu().numpy().view(np.int64), y3.cpu().numpy().view(np.int64), z3.cpu().numpy()], axis=1)
    gk = key.cpu().numpy()[first.cpu().numpy().view(np.uint32)]
    gathered = [None] * world
    dist.all_gather_object(gathered, (rows, gk, cnt.cpu().numpy()))
    if rank == 0:
        L, R = orc.join_pairs(ga, None, gb, None)
        P, Q = orc.join_pairs(ga[L], None, gc, None)
        want = np.stack([ga[L][P], gx.view(np.int64)[L][P], gy.view(np.int64)[R][P], gz[Q]], axis=1)
        got = np.concatenate([g[0] for g in gathered])
        assert got.shape == want.shape, (got.shape, want.shape)
        assert np.array_equal(got[np.lexsort(got.T[::-1])], want[np.lexsort(want.T[::-1])])
        ef, ec = orc.group_count(ga[L][P], None)
        ek = ga[L][P][ef]
        k2, c2 = np.concatenate([g[1] for g in gathered]), np.concatenate([g[2] for g in gathered])
        o1, o2 = np.argsort(k2, kind="stable"), np.argsort(ek, kind="stable")
        assert np.array_equal(k2[o1], ek[o2]) and np.array_equal(c2[o1], ec[o2])
        print("rccl table shuffle payload join ok", len(got), "joined rows", len(ek), "groups", "world", world)
    # ---- the north-star pipeline itself (bench.py --gpus N): partition by destination, key exchange, split join operator
    #      with the DEVICE operators: 8-byte and 4-byte wire format (int32 keys are consumed as they arrive), one and two
    #      pieces per table, narrow and 64-bit form of the join
    m = 1_300_000
    tot = m * world
    rng = np.random.default_rng(23)
    fa = rng.permutation(tot).astype(np.int64) - tot // 3
    fb = rng.integers(-tot // 3, tot - tot // 3, tot, dtype=np.int64)
    sl = slice(rank * m, (rank + 1) * m)
    da, db = dev.to_dev(fa[sl]), dev.to_dev(fb[sl])
    ek, ec, _, ej = orc.join_group_count(fa, None, fb, None)
    for wire32, chunks, narrow in ((False, 1, 1), (True, 1, 1), (True, 2, 1), (True, 1, 0), (False, 2, 0)):
        dev.set_narrow_keys(narrow)
        pipe = DistributedJoinGroupCount(dev, world, rank, m, chunks=chunks, wire32=wire32)
        g, j = pipe.run(da, db)
        k, c, _ = pipe.last
        gathered = [None] * world
        dist.all_gather_object(gathered, (k.cpu().numpy(), c.cpu().numpy(), int(j)))
        if rank == 0:
            k2, c2 = np.concatenate([x[0] for x in gathered]), np.concatenate([x[1] for x in gathered])
            assert sum(x[2] for x in gathered) == ej
            o1, o2 = np.argsort(k2, kind="stable"), np.argsort(ek, kind="stable")
            assert np.array_equal(k2[o1], ek[o2]) and np.array_equal(c2[o1], ec[o2]), (wire32, chunks, narrow)
        if world == 1:      # one rank: the result keeps the reference's first-occurrence order of what was received
            assert k.numel() == len(ek)
    dev.set_narrow_keys(1)
    if rank == 0:
        print("rccl distributed join group count ok", len(ek), "groups", ej, "joined rows")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
