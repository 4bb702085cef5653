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
from midoridb_amd.shuffle import DistributedJoinGroupCount  # noqa: E402


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

    pipe = DistributedJoinGroupCount(None, world, rank, n, partition_fn=partition_fn, join_fn=join_fn, device=torch.device("cpu"))
    g, j = pipe.run(torch.from_numpy(a), torch.from_numpy(b), None)
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
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
