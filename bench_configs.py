"""bench_configs.py - the other BASELINE.json configurations behind `bench.py --config 4|5` (C3, the
north-star query, is bench.py itself).  Same contract: tables resident in HBM when the timed region starts, W untimed warm-up
steps, exactly K timed steps between barriers, MAX over ranks, one JSON line from rank 0.

  --config 4  BASELINE configs[3]: `SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b` over two key columns, unique keys on both
              sides (10^9 rows per table over 8 GPUs = 1.25e8 rows per table per GPU).  N > 1 (or --force-shuffle):
              mdb_dist_join_pairs() - both tables hash-partitioned by destination, exchanged over RCCL, joined locally, the
              joined rows' key column materialised; N = 1: mdb_dev_join_pairs() + the projection gather.
  --config 5  BASELINE configs[4]: A(id_a, x DOUBLE) JOIN B(id_b, y DOUBLE) JOIN C(id_c, z INT) on one key + GROUP BY id_a
              COUNT(*), through query_execute() on device-resident tables with results kept on the device
              (mdb_database_results_on_device), sharded through mdb_database_set_dist() for N > 1; the join-only form
              `SELECT *` (x, y, z carried as payload) is timed beside it.
"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0
METRIC = "joined rows/sec, 2x10^8-row INT64 INNER JOIN+GROUP BY, 1/2/4/8 MI355X"


def run(args, world, rank, local_rank, json_fd, watchdog=None):
    from midoridb_amd.dev import DeviceCtx
    from midoridb_amd.dist import DatabaseDevice, DistCtx
    use_dist = world > 1 or args.force_shuffle
    host_wire = getattr(args, "transport", "rccl") == "test"     # N ranks on GPU 0, blocks through host memory (bench.py --transport test)
    n = args.rows
    total = n * world

    def beat(what):
        if watchdog:
            watchdog.beat(what)

    def all_reduce_(t, op):
        if host_wire:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    def make_dist(dev):
        dx = DistCtx.over_host_group(dev) if host_wire else DistCtx.from_torch(dev)
        seen = dx.allreduce_sum([1])[0]
        if seen != world:
            raise SystemExit(f"[bench] rank {rank}: the library's communicator saw {seen} ranks, the launcher {world}")
        return dx

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, warmup):
        r = None
        for _ in range(max(warmup, 1)):
            beat("warm-up step")
            r = step()
        beat("timed steps")
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = step()
        barrier()
        dt = time.perf_counter() - t0
        red = torch.tensor([dt, float(r)], dtype=torch.float64, device=f"cuda:{local_rank}")
        if use_dist:
            tmax = red[:1].clone()
            all_reduce_(tmax, dist.ReduceOp.MAX)
            all_reduce_(red, dist.ReduceOp.SUM)
            red[0] = tmax[0]
        beat("timed steps done")
        return float(red[0].item()), int(red[1].item())

    line = {"metric": METRIC, "unit": "joined rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic"}
    if args.config == 4:
        dev = DeviceCtx(local_rank)
        dx = make_dist(dev) if use_dist else None
        a = dev.gen_keys(n, rank * n, total, 42, 0)
        b = dev.gen_keys(n, rank * n, total, 43, 0)
        if dx is not None:
            # catalog statistics (computed once per table, outside the timed region): the two tables' GLOBAL key ranges
            from midoridb_amd.dist import WIRE_32
            rng = []
            for col in (a, b):
                lo, hi = dev.key_range(col)
                t = torch.tensor([-lo, hi], dtype=torch.float64, device=dev.device)
                all_reduce_(t, dist.ReduceOp.MAX)
                rng.append((-int(t[0].item()), int(t[1].item())))
            dx.set_wire(WIRE_32)
            dx.set_key_ranges(rng[0], rng[1])

        def step():
            if dx is not None:
                key, _, _, J = dx.join_pairs(a, None, [], b, None, [])
                return J
            if not args.reference_order:
                return dev.join_keys(a, None, b, None).numel()      # the key column of every joined row, any order (mdb_dev_join_keys)
            # the reference's left-major row order: a primary-key join through the ordered join + GROUP BY + COUNT(*) operator
            # (mdb_dev_join_keys_ordered, round 4); with duplicates on a side the pairs + the key gather
            k = dev.join_keys_ordered(a, None, b, None)
            if k is not None:
                return k.numel()
            pl, pr = dev.join_pairs(a, None, b, None)
            J = pl.numel()
            dev.gather64(a, None, pl, J)
            return J
        dt, joined = timed(step, args.steps, args.warmup)
        fused = dx is not None and dx.last_fused()
        dev.prof_enable(True)
        dev.prof_reset()
        for _ in range(3):
            step()
        prof = dev.prof_read()
        dev.prof_enable(False)
        algo = 8 * 2 * total + 8 * joined * 2
        kern = {k: {"launches_per_step": v[0] / 3, "ms_per_step": v[1] / 3} for k, v in prof.items()}
        dom = max(kern, key=lambda k: kern[k]["ms_per_step"]) if kern else None
        line.update({
            "value": joined / (dt / args.steps), "ms_per_step": dt / args.steps * 1e3,
            "config": {"workload": f"BASELINE configs[3] (SURVEY C4): SELECT * FROM A INNER JOIN B ON id_a = id_b, key columns only, unique keys, "
                                   f"{n} rows/table/GPU x {world} GPU = {total} rows/table; the joined rows' key column is materialised once "
                                   "(id_a and id_b hold the same value in every joined row)",
                       "rows_per_table_per_gpu": n, "rows_per_table_total": total, "joined_rows": joined,
                       "parallelism": (f"hash-partition x{world}, mdb_dist_join_pairs"
                                       + (" (key columns only: first-level partition regions on the wire, every key written COUNT times)"
                                          if fused else " (RCCL all-to-all of keys by destination, local join)")
                                       + (" (forced shuffle)" if world == 1 else "")) if use_dist else
                                      ("single GPU, mdb_dev_join_keys_ordered (the reference's left-major row order; a primary-key join: the ordered join + GROUP BY + COUNT(*) operator, J == G)" if args.reference_order else
                                       "single GPU, mdb_dev_join_keys (key column of the joined rows in unspecified order, as the sharded "
                                       "form delivers it: regions of 2-byte words, no row ids)")},
            "pipeline": {"algorithmic_bytes": algo, "achieved_GBs": algo / (dt / args.steps) / 1e9,
                         "frac_of_peak": algo / (dt / args.steps) / 1e9 / HBM_PEAK_GBS / world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": algo / world / (dt / args.steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": algo / world / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "note": "whole step on SURVEY 8d's algorithmic bytes per GPU (8 B per key read + 16 B per joined row written); "
                                 "per-kernel times under `kernels`"},
            "kernels": kern,
            "cpu_baseline": None,
        })
        if dx is not None:
            dx.close()
        dev.close()
    else:
        from midoridb_amd.query import DB
        GROUPED = ("SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c GROUP BY id_a;")
        JOINED = "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c;"
        os.environ["MIDORIDB_DEVICE"] = str(local_rank)
        with DB() as db:
            if use_dist:
                make_dist(DatabaseDevice(db, local_rank)).attach_to_database(db)
            db.execute("CREATE TABLE A (id_a INT, x DOUBLE);")
            db.execute("CREATE TABLE B (id_b INT, y DOUBLE);")
            db.execute("CREATE TABLE C (id_c INT, z INT);")
            for t, seed in (("A", 42), ("B", 43), ("C", 44)):
                db.generate_shard(t, n, rank * n, total, seed, [0, 0])
            db.results_on_device(True)
            res = {}
            for name, sql, steps in (("grouped", GROUPED, args.steps), ("joined", JOINED, max(2, args.steps // 2))):
                def step(sql=sql):
                    r = db.query_device(sql, copy=False)
                    step.rows = r[3]
                    return r[4]
                dt, joined = timed(step, steps, args.warmup)
                res[name] = {"ms_per_step": dt / steps * 1e3, "value": joined / (dt / steps), "joined_rows": joined, "result_rows_rank0": step.rows,
                             "steps": steps}
            g = res["grouped"]
            algo = 8 * 3 * total + 16 * g["joined_rows"]	# three key columns read once, (key, COUNT) per group written (G = joined rows here)
            algo_j = 8 * 6 * total + 8 * 6 * res["joined"]["joined_rows"]
            res["joined"]["pipeline"] = {"algorithmic_bytes": algo_j, "frac_of_peak": algo_j / (res["joined"]["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS / world}
            line.update({
                "value": g["value"], "ms_per_step": g["ms_per_step"], "steps": g["steps"],
                "config": {"workload": f"BASELINE configs[4] (SURVEY C5): A(id_a, x DOUBLE) JOIN B(id_b, y DOUBLE) JOIN C(id_c, z INT) on id_a "
                                       f"GROUP BY id_a COUNT(*), unique keys, {n} rows/table/GPU x {world} GPU, through query_execute() with "
                                       "results kept on the device",
                           "query": GROUPED, "rows_per_table_per_gpu": n, "rows_per_table_total": total, "joined_rows": g["joined_rows"],
                           "parallelism": (f"hash-partition x{world}, query_execute() in sharded mode (mdb_database_set_dist, RCCL)"
                                           + (" (forced shuffle)" if world == 1 else "")) if use_dist else "single GPU"},
                "pipeline": {"algorithmic_bytes": algo, "achieved_GBs": algo / (g["ms_per_step"] * 1e-3) / 1e9,
                             "frac_of_peak": algo / (g["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS / world},
                "roofline": {"bound": "hbm", "kernel": "whole statement", "achieved": algo / world / (g["ms_per_step"] * 1e-3) / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo / world / (g["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None},
                "join_only_form": dict(res["joined"], query=JOINED, note="x, y DOUBLE and z INT carried as payload: 6 result columns"),
                "cpu_baseline": None,
            })
    if use_dist:    # (make_dist() has checked that the library's own communicator saw every rank)
        line["config"]["rccl_ranks_seen"] = "test transport" if host_wire else world
        if host_wire:
            line["config"]["transport"] = "host memory through a gloo process group, all ranks on GPU 0 (--transport test)"
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
