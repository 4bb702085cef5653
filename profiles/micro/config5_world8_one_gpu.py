"""BASELINE configs[4] with all EIGHT ranks on one GPU: A(id_a, x DOUBLE) JOIN B(id_b, y DOUBLE) JOIN C(id_c, z INT) on one key + GROUP BY
id_a COUNT(*), and its join-only form (x, y, z carried), through query_execute() in sharded mode (mdb_database_set_dist; the blocks moved
through host memory by the test transport - the pool hands out one-GPU boxes).  Not a timing; the whole sharded executor at the world and
size the configuration names: 10^8 rows per table over 8 ranks by default (argument: rows per rank).

Checks: the ranks' groups add up to the table size with COUNT(*) = 1 each (unique keys, every key in all three tables) and every key exactly
once; the join-only form returns every row of A once, with ITS x, y and z: the checksums (sums of the 8-byte cells mod 2^64) of the result's
x, y, z columns over all ranks equal those of the base tables' columns over all ranks.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29735 profiles/micro/config5_world8_one_gpu.py [rows per rank] [grouped-only]
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
from midoridb_amd.query import DB  # noqa: E402
from midoridb_amd.dist import DatabaseDevice  # noqa: E402
from _dist_gpu_worker import gloo_transport  # noqa: E402

M = 2**64 - 1


def csum(t):
    return int(t.view(torch.int64).sum().item()) & M if t is not None and t.numel() else 0


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
    grouped_only = len(sys.argv) > 2
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = n * world
    out = {"workload": f"BASELINE configs[4]: three tables of {total} unique keys over {world} ranks on one GPU (test transport)", "rows_per_rank": n}
    with DB() as db:
        gloo_transport(DatabaseDevice(db, 0), world, rank).attach_to_database(db)
        db.execute("CREATE TABLE A (id_a INT, x DOUBLE);")
        db.execute("CREATE TABLE B (id_b INT, y DOUBLE);")
        db.execute("CREATE TABLE C (id_c INT, z INT);")
        for t, seed in (("A", 42), ("B", 43), ("C", 44)):
            db.generate_shard(t, n, rank * n, total, seed, [0, 0])
        db.results_on_device(True)

        def gather(vals):
            parts = [None] * world
            dist.all_gather_object(parts, vals)
            return [sum(p[i] for p in parts) for i in range(len(vals))]
        t0 = time.perf_counter()
        names, types, cols, nrows, joined, _ = db.query_device("SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c GROUP BY id_a;")
        dt = time.perf_counter() - t0
        k, c = cols[names.index("A.id_a")], cols[names.index("COUNT(*)")]
        g, cs, ks, k2 = gather([nrows, csum(c), csum(k), int((k * k).sum().item()) & M if nrows else 0])
        ok_g = g == total and cs == total and (ks & M) == (total * (total - 1) // 2) & M and (k2 & M) == ((total - 1) * total * (2 * total - 1) // 6) & M
        out["grouped"] = {"groups_total": g, "sum_of_counts": cs, "every_key_exactly_once_with_count_1": bool(ok_g), "call_seconds_rank0": round(dt, 2)}
        del cols, k, c
        if not grouped_only:
            base = []
            for t, col in (("A", "x"), ("B", "y"), ("C", "z")):
                r = db.query_device(f"SELECT {col} FROM {t};")
                base.append(csum(r[2][0]))
                del r
            base = gather(base)
            t0 = time.perf_counter()
            names, types, cols, nrows, joined, _ = db.query_device("SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c;")
            dt = time.perf_counter() - t0
            res = gather([nrows] + [csum(cols[names.index(nm)]) for nm in ("A.x", "B.y", "C.z", "A.id_a", "B.id_b", "C.id_c")])
            keysum = (total * (total - 1) // 2) & M
            ok_j = res[0] == total and all((res[1 + i] & M) == (base[i] & M) for i in range(3)) and all((res[4 + i] & M) == keysum for i in range(3))
            out["join_only"] = {"rows_total": res[0], "payload_checksums_equal_the_tables": bool(ok_j), "call_seconds_rank0": round(dt, 2)}
            assert ok_j, (res, base)
        assert ok_g, (g, cs, ks, k2)
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
