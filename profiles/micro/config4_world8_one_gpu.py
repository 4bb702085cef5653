"""BASELINE configs[3] at FULL size - SELECT * FROM A INNER JOIN B ON id_a = id_b over two key columns, 10^9 unique keys per table,
hash-partitioned over 8 ranks - through the product's C path (mdb_dist_join_pairs, keys only: first-level regions on the wire, the plan of a
2^30-value window: 512 digits, the receiver's own 8-bit level, leaves of 2^13 values) with all eight ranks on ONE GPU: eight contexts on one
device, the blocks moved through host memory by the test transport of tests/_dist_gpu_worker.py (the pool hands out one-GPU boxes).  Not a
timing - the transport is gloo - but the whole exchange logic at the size and world the configuration names.

Checks (size-independent properties): the ranks' joined rows add up to 10^9; every key lands on exactly one rank (the sum of all returned keys
is n (n - 1) / 2 for a permutation of 0 .. n - 1, and so is the sum of their squares mod 2^64); each rank's keys are the ones that hash to it.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29733 profiles/micro/config4_world8_one_gpu.py [rows per rank]
"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from midoridb_amd.dev import DeviceCtx  # noqa: E402
from midoridb_amd.dist import WIRE_32  # noqa: E402
from _dist_gpu_worker import gloo_transport  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = DeviceCtx(0)
    dx = gloo_transport(dev, world, rank)
    total = n * world
    a = dev.gen_keys(n, rank * n, total, 42, 0)
    b = dev.gen_keys(n, rank * n, total, 43, 0)
    dx.set_wire(WIRE_32)
    dx.set_key_ranges((0, total - 1), (0, total - 1))
    t0 = time.perf_counter()
    key, _, _, J = dx.join_pairs(a, None, [], b, None, [])
    dt = time.perf_counter() - t0
    assert dx.last_fused(), "expected the regions-on-the-wire path"
    plan = dx.last_plan()
    s1 = int(key.sum().item()) & (2**64 - 1)					# (wraps like the expected values below)
    s2 = int((key * key).sum().item()) & (2**64 - 1)
    # the keys of this rank are the ones whose window hash puts them here: spot-check a sample on the host
    from oracle import np_oracle as orc
    sample = key[:: max(1, key.numel() // 200_000)].cpu().numpy()
    assert bool((orc.dest_of_fused(sample, world, 0, total) == rank).all()), "a key on the wrong rank"
    parts = [None] * world
    dist.all_gather_object(parts, (J, s1, s2))		# (python integers: the checksums wrap at 2^64, a transport's counters need not)
    tot = [sum(p[i] for p in parts) for i in range(3)]
    exp1 = (total * (total - 1) // 2) & (2**64 - 1)
    exp2 = ((total - 1) * total * (2 * total - 1) // 6) & (2**64 - 1)
    ok = tot[0] == total and (tot[1] & (2**64 - 1)) == exp1 and (tot[2] & (2**64 - 1)) == exp2
    if rank == 0:
        print(json.dumps({"workload": f"BASELINE configs[3]: {total} unique keys per table over {world} ranks on one GPU (test transport)",
                          "rows_per_rank": n, "joined_rows_total": tot[0], "plan": plan, "every_key_exactly_once": bool(ok),
                          "call_seconds_rank0_through_host_memory": round(dt, 2)}), flush=True)
    assert ok, (tot, exp1, exp2)
    dx.close()
    dev.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
