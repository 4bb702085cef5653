"""The north-star query (BASELINE configs[2], variant D) at the 8-GPU weak-scaling size - 10^8 rows per table per rank, 8 x 10^8 per table -
with all eight ranks on ONE GPU: mdb_dist_join_group_count through the test transport (host memory).  Not a timing; the plan bench.py --gpus 8
will run (4096 first-level digits, 512 per rank, 64 segments per leaf, 2-byte words) at full size.  Checks: 5 x 10^7 groups in all, each with
COUNT(*) = 16, every key on exactly one rank, 8 x 10^8 joined rows.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29736 profiles/micro/north_star_world8_one_gpu.py [rows per rank]
"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from midoridb_amd.dev import DeviceCtx  # noqa: E402
from midoridb_amd.dist import WIRE_32  # noqa: E402
from _dist_gpu_worker import gloo_transport  # noqa: E402

M = 2**64 - 1


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = DeviceCtx(0)
    dx = gloo_transport(dev, world, rank)
    total = n * world
    a = dev.gen_keys(n, rank * n, total, 42, 0)
    b = dev.gen_keys(n, rank * n, total, 43, total // 16)
    dx.set_wire(WIRE_32)
    dx.set_key_ranges((0, total - 1), (0, total // 16 - 1))
    cap = int(total // 16 // world * 1.5) + 65536
    out = (torch.empty(cap, dtype=torch.int64, device=dev.device), torch.empty(cap, dtype=torch.int64, device=dev.device))
    t0 = time.perf_counter()
    k, c, j = dx.join_group_count(a, None, b, None, out=out)
    dt = time.perf_counter() - t0
    assert dx.last_fused(), "expected the regions-on-the-wire path"
    parts = [None] * world
    dist.all_gather_object(parts, (k.numel(), j, int(k.sum().item()) & M, int(c.min().item()), int(c.max().item())))
    G, J, ks = sum(p[0] for p in parts), sum(p[1] for p in parts), sum(p[2] for p in parts) & M
    g = total // 16
    ok = G == g and J == total and ks == (g * (g - 1) // 2) & M and all(p[3] == 16 and p[4] == 16 for p in parts)
    if rank == 0:
        print(json.dumps({"workload": f"north-star query, variant D, {n} rows per table per rank x {world} ranks on one GPU (test transport)",
                          "groups_total": G, "joined_rows_total": J, "plan": dx.last_plan(), "every_key_once_with_count_16": bool(ok),
                          "call_seconds_rank0_through_host_memory": round(dt, 2)}), flush=True)
    assert ok, parts
    dx.close()
    dev.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
