"""bench_configs.py - the other BASELINE.json configurations behind `bench.py --config 2|4|5` (C3, the
north-star query, is bench.py itself).  Same contract: tables resident in HBM when the timed region starts, W untimed warm-up
steps, exactly K timed steps between barriers, MAX over ranks, one JSON line from rank 0.

  --config 2  BASELINE configs[1]: `SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b` over A(id_a, fa), B(id_b, fb), unique keys,
              10^7 rows per table (--rows; one GPU), through query_execute() with results kept on the device: the join that
              carries the right table's payload (mdb_dev_join_payload).
  --config 4  BASELINE configs[3]: `SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b` over two key columns, unique keys on both
              sides (10^9 rows per table over 8 GPUs = 1.25e8 rows per table per GPU).  N > 1 (or --force-shuffle):
              mdb_dist_join_pairs() - both tables hash-partitioned by destination, exchanged over RCCL, joined locally, the
              joined rows' key column materialised; N = 1: mdb_dev_join_keys() / --reference-order: mdb_dev_join_keys_ordered().
  --config 5  BASELINE configs[4]: A(id_a, x DOUBLE) JOIN B(id_b, y DOUBLE) JOIN C(id_c, z INT) on one key + GROUP BY id_a
              COUNT(*), through query_execute() on device-resident tables with results kept on the device
              (mdb_database_results_on_device), sharded through mdb_database_set_dist() for N > 1; the join-only form
              `SELECT *` (x, y, z carried as payload) is timed beside it.

Every line carries, like the north-star line: `kernels` (live HIP-event time per profiler name), `roofline.traffic` (HBM bytes per step from
the committed rocprofv3 PMC summary of the same command, profiles/rNN/rocprof_summary_config<N>.json: bytes per launch x launches per step),
`cpu_baseline` (the REAL reference executor, oracle/_ref, on the same statement at the largest size it finishes in seconds - its join is a
nested loop -, N = 1 only) and `cpu_check` (a CPU computation of the same result at the FULL size whose outcome is compared with the GPU's).
"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0
METRIC = "joined rows/sec, 2x10^8-row INT64 INNER JOIN+GROUP BY, 1/2/4/8 MI355X"
ROOT = os.path.dirname(os.path.abspath(__file__))
PROFILE_ROUNDS = ("r06", "r05", "r04")


def _pmc_kernels(tag):
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", rnd, f"rocprof_summary_{tag}.json")
        try:
            with open(path) as f:
                return json.load(f)["kernels"], f"profiles/{rnd}/rocprof_summary_{tag}.json"
        except Exception:
            continue
    return None, None


def pmc_step_traffic(kern, tag, algo_bytes):
    """HBM bytes one step moves by the PMC counters of the committed summary of the same command: per profiler name, bytes per launch
    (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE) x the launches per step measured live; None when no summary is committed."""
    ks, src = _pmc_kernels(tag)
    if ks is None:
        return None
    total, missing = 0.0, []
    for name, d in kern.items():
        num = den = 0.0
        for rn in d.get("rocprof_names", []):
            k = ks.get(rn)
            if k and "hbm_read_bytes" in k and "hbm_write_bytes" in k:
                w = float(k.get("calls", 1) or 1)
                num += w * (k["hbm_read_bytes"] + k["hbm_write_bytes"])
                den += w
        if den:
            total += num / den * d["launches_per_step"]
        else:
            missing.append(name)
    return {"pmc_bytes_per_step": total, "algorithmic_bytes": algo_bytes, "traffic_over_algorithmic": total / algo_bytes if algo_bytes else None,
            "source": src, "kernels_without_counters": missing}


class Prof:
    """live per-kernel profile of a raw device context handle (a DeviceCtx's or a database's)"""

    def __init__(self, lib, handle):
        from midoridb_amd.dev import _bind, ProfEntry
        _bind(lib)
        self.lib, self.h, self.Entry = lib, handle, ProfEntry

    def run(self, step, reps=3):
        self.lib.mdb_dev_prof_enable(self.h, 1)
        self.lib.mdb_dev_prof_reset(self.h)
        for _ in range(reps):
            step()
        self.lib.mdb_dev_sync(self.h)
        buf = (self.Entry * 96)()
        cnt = ctypes.c_int()
        self.lib.mdb_dev_prof_read(self.h, buf, 96, ctypes.byref(cnt))
        kern = {}
        for i in range(cnt.value):
            if not buf[i].launches:
                continue
            name = buf[i].name.decode()
            sym = ctypes.create_string_buffer(16384)
            self.lib.mdb_dev_prof_symbols(self.h, name.encode(), sym, len(sym))
            kern[name] = {"launches_per_step": buf[i].launches / reps, "ms_per_step": buf[i].total_ms / reps,
                          "rocprof_names": [x for x in sym.value.decode().split("\n") if x]}
        self.lib.mdb_dev_prof_enable(self.h, 0)
        return kern


def reference_baseline(create, tables, sql, n, what):
    """the REAL reference executor (oracle/_ref) on `sql` over tables of n and n // 2 rows (its join is a nested loop: quadratic) -> the
    cpu_baseline object, or a note why there is none"""
    try:
        from oracle import ref as refmod
        if not refmod.available():
            return {"value": None, "note": "oracle/_ref is not built on this box"}
        times, joined = {}, 0
        for m in (n // 2, n):
            db = refmod.RefDB()
            rng = np.random.default_rng(42)
            for name, cols in create:
                db.create_int_table(name, cols)
            for name, ncols in tables:
                key = rng.permutation(np.arange(m, dtype=np.int64))
                db.bulk_insert(name, [key] + [key * (3 + i) for i in range(ncols - 1)])
            t0 = time.perf_counter()
            cols, rows = db.query(sql)
            times[m] = time.perf_counter() - t0
            joined = len(rows)
            db.close()
        dt = times[n]
        return {"value": n / dt, "unit": "joined rows/s", "cores": 1, "kind": "reference",
                "sample": f"{what}, {n} rows per table (unique keys, INT columns), real reference executor via oracle/_ref: {dt:.2f} s "
                          f"({n // 2} rows: {times[n // 2]:.2f} s - a nested loop, O(nA*nB)); {joined} result rows"}
    except Exception as e:  # pragma: no cover - diagnostic path
        return {"value": None, "note": f"reference baseline failed: {e}"}


def run(args, world, rank, local_rank, json_fd, watchdog=None):
    from midoridb_amd.dev import DeviceCtx
    from midoridb_amd.dist import DatabaseDevice, DistCtx
    use_dist = world > 1 or args.force_shuffle
    host_wire = getattr(args, "transport", "rccl") == "test"     # N ranks on GPU 0, blocks through host memory (bench.py --transport test)
    full_rows = 10_000_000 if args.config == 2 else 100_000_000
    n = args.rows if args.rows else full_rows
    total = n * world
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline

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

    last_dist = {}

    def timed(step, steps, warmup, dev=None):
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
        # how the steps are distributed: the same number again, each between its own synchronisations; `value` stays the mean of the loop above
        c0 = dev.counters() if dev is not None else None
        each = []
        for _ in range(steps):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            each.append((time.perf_counter() - t1) * 1e3)
        v = sorted(each)
        last_dist.clear()
        last_dist.update({"steps": steps, "min": v[0], "median": v[steps // 2], "p90": v[min(steps - 1, (9 * steps) // 10)], "max": v[-1], "mean": sum(v) / steps})
        if dev is not None:
            c1 = dev.counters()
            last_dist.update(retries=c1["retries"] - c0["retries"], arena_grows=c1["arena_grows"] - c0["arena_grows"],
                             alloc_misses=c1["alloc_misses"] - c0["alloc_misses"], plan=dev.last_plan())
        return float(red[0].item()), int(red[1].item())

    def roof(algo, ms, kern, tag, note):
        dom = max(kern, key=lambda k: kern[k]["ms_per_step"]) if kern else None
        tr = pmc_step_traffic(kern, tag, algo) if (world == 1 and not use_dist and n == full_rows) else None
        return {"bound": "hbm", "kernel": "whole statement", "longest_kernel": dom, "achieved": algo / world / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": algo / world / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": tr["pmc_bytes_per_step"] if tr else None, "traffic_over_algorithmic": tr["traffic_over_algorithmic"] if tr else None,
                "traffic_source": tr["source"] if tr else None, "kernels_without_counters": tr["kernels_without_counters"] if tr else None,
                "note": note}

    line = {"metric": METRIC, "unit": "joined rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic"}
    if args.config == 4:
        dev = DeviceCtx(local_rank)
        dx = make_dist(dev) if use_dist else None
        a = dev.gen_keys(n, rank * n, total, 42, 0)
        b = dev.gen_keys(n, rank * n, total, 43, 0)
        stats = (dev.key_range(a), dev.key_range(b))    # catalog statistics: computed once per table, outside the timed region
        if dx is not None:
            from midoridb_amd.dist import WIRE_32
            rng = []
            for lo, hi in stats:
                t = torch.tensor([-lo, hi], dtype=torch.float64, device=dev.device)
                all_reduce_(t, dist.ReduceOp.MAX)
                rng.append((-int(t[0].item()), int(t[1].item())))
            dx.set_wire(WIRE_32)
            dx.set_key_ranges(rng[0], rng[1])
        last = {}

        def step():
            if dx is not None:
                key, _, _, J = dx.join_pairs(a, None, [], b, None, [])
                last["keys"] = key
                return J
            dev.call_stats(a, stats[0], b, stats[1])
            try:
                if not args.reference_order:
                    last["keys"] = dev.join_keys(a, None, b, None)      # the key column of every joined row, any order (mdb_dev_join_keys)
                    return last["keys"].numel()
                # the reference's left-major row order: a primary-key join through the ordered join + GROUP BY + COUNT(*) operator
                # (mdb_dev_join_keys_ordered); with duplicates on a side the pairs + the key gather
                k = dev.join_keys_ordered(a, None, b, None)
                if k is not None:
                    last["keys"] = k
                    return k.numel()
                pl, pr = dev.join_pairs(a, None, b, None)
                J = pl.numel()
                last["keys"] = dev.gather64(a, None, pl, J)
                return J
            finally:
                dev.call_stats()
        dt, joined = timed(step, args.steps, args.warmup, dev)
        ms = dt / args.steps * 1e3
        fused = dx is not None and dx.last_fused()
        kern = Prof(dev.lib, dev.h).run(step)
        algo = 8 * 2 * total + 8 * joined * 2
        written = 8 * joined        # ONE key column is written: id_a and id_b hold the same value in every joined row
        line.update({
            "value": joined / (dt / args.steps), "ms_per_step": ms, "step_ms": dict(last_dist),
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
                         "frac_of_peak": algo / (dt / args.steps) / 1e9 / HBM_PEAK_GBS / world,
                         "bytes_actually_written": written,
                         "note": "algorithmic_bytes is SURVEY 8d's formula (8 B per key read + 16 B per joined row: BOTH key columns of SELECT *); "
                                 "the operator writes ONE 8-byte column for the two (they are equal in every joined row): bytes_actually_written"},
            "roofline": roof(algo, ms, kern, "config4" + ("_reference_order" if args.reference_order else ""),
                             "whole step on SURVEY 8d's algorithmic bytes per GPU; per-kernel times under `kernels`"),
            "kernels": kern,
        })
        if want_cpu:
            beat("CPU baseline (rank 0)")
            line["cpu_baseline"] = reference_baseline([("A", ["id_a"]), ("B", ["id_b"])], [("A", 1), ("B", 1)],
                                                      "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b;", 5000, "BASELINE configs[3]'s statement")
            beat("CPU check (rank 0)")
            try:        # the same join on all host cores, full size: every key of A that occurs in B, once (unique keys)
                from oracle import cpu
                t0 = time.perf_counter()
                ek, ec, _, ej = cpu.hash_join_group_count(a.cpu().numpy(), None, b.cpu().numpy(), None, os.cpu_count() or 1)
                sec = time.perf_counter() - t0
                got = np.sort(last["keys"].cpu().numpy())
                line["cpu_check"] = {"what": "oracle/cpu_hash.c (multi-threaded hash join, pinned to the reference's vectors) on the benchmark's own tables, full size",
                                     "cores": os.cpu_count() or 1, "seconds": sec, "value": ej / sec, "unit": "joined rows/s",
                                     "gpu_result_identical": bool(ej == joined and bool((ec == 1).all()) and np.array_equal(got, np.sort(ek)))}
            except Exception as e:  # pragma: no cover
                line["cpu_check"] = {"error": str(e)}
        else:
            line["cpu_baseline"] = None
        if dx is not None:
            dx.close()
        dev.close()
    elif args.config == 2:
        from midoridb_amd.query import DB
        SQL = "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b;"
        os.environ["MIDORIDB_DEVICE"] = str(local_rank)
        with DB() as db:
            db.execute("CREATE TABLE A (id_a INT, fa INT);")
            db.execute("CREATE TABLE B (id_b INT, fb INT);")
            db.generate_shard("A", n, 0, n, 42, [0, 0])     # id_a = perm_42(i), fa = perm_43(i)
            db.generate_shard("B", n, 0, n, 52, [0, 0])     # id_b = perm_52(i), fb = perm_53(i)
            db.results_on_device(True)
            last = {}

            def step():
                r = db.query_device(SQL, copy=False)
                last["rows"] = r[3]
                return r[4] if r[4] else r[3]
            dt, joined = timed(step, args.steps, args.warmup, db)
            ms = dt / args.steps * 1e3
            kern = Prof(db.lib, db.device_handle()).run(step)
            plan = db.last_plan()
            algo = 8 * 4 * n + 8 * 4 * joined        # four 8-byte columns read once, four written per joined row
            line.update({
                "value": joined / (dt / args.steps), "ms_per_step": ms, "step_ms": dict(last_dist),
                "config": {"workload": f"BASELINE configs[1] (SURVEY C2): SELECT * FROM A(id_a, fa) INNER JOIN B(id_b, fb) ON id_a = id_b, unique keys, "
                                       f"{n} rows per table, one GPU, through query_execute() with results kept on the device",
                           "query": SQL, "rows_per_table_per_gpu": n, "rows_per_table_total": n, "joined_rows": joined,
                           "parallelism": "single GPU", "plan": plan},
                "pipeline": {"algorithmic_bytes": algo, "achieved_GBs": algo / (ms * 1e-3) / 1e9, "frac_of_peak": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "roofline": roof(algo, ms, kern, "config2", "whole statement on SURVEY 8d's bytes (every column read once, every result cell written once)"),
                "kernels": kern,
            })
            if want_cpu:
                beat("CPU baseline (rank 0)")
                line["cpu_baseline"] = reference_baseline([("A", ["id_a", "fa"]), ("B", ["id_b", "fb"])], [("A", 2), ("B", 2)], SQL, 5000,
                                                          "BASELINE configs[1]'s statement")
                beat("CPU check (rank 0)")
                try:    # numpy on the host, full size: B's payload of every A row's partner (unique keys: a sort + a search)
                    r = db.query_device(SQL, copy=True)
                    cols = dict(zip(r[0], r[2]))
                    ka, fa = cols["A.id_a"].cpu().numpy(), cols["A.fa"].cpu().numpy()
                    kb, fb = cols["B.id_b"].cpu().numpy(), cols["B.fb"].cpu().numpy()
                    a_id, a_f = (db.query_device(f"SELECT {c} FROM A;", copy=True)[2][0].cpu().numpy() for c in ("id_a", "fa"))
                    b_id, b_f = (db.query_device(f"SELECT {c} FROM B;", copy=True)[2][0].cpu().numpy() for c in ("id_b", "fb"))
                    t0 = time.perf_counter()
                    order = np.argsort(b_id, kind="stable")
                    pos = np.searchsorted(b_id[order], a_id)
                    hit = (pos < len(b_id)) & (b_id[order][np.minimum(pos, len(b_id) - 1)] == a_id)
                    exp_fb = b_f[order][pos[hit]]
                    sec = time.perf_counter() - t0
                    line["cpu_check"] = {"what": "numpy sort + binary search on the host over the benchmark's own tables, full size (one thread)", "cores": 1,
                                         "seconds": sec, "value": int(hit.sum()) / sec, "unit": "joined rows/s",
                                         "gpu_result_identical": bool(np.array_equal(ka, a_id[hit]) and np.array_equal(fa, a_f[hit]) and
                                                                      np.array_equal(kb, a_id[hit]) and np.array_equal(fb, exp_fb))}
                except Exception as e:  # pragma: no cover
                    line["cpu_check"] = {"error": str(e)}
            else:
                line["cpu_baseline"] = None
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
            prof = Prof(db.lib, db.device_handle())
            # The three key columns are complete primary keys over one range: the catalog KNOWS that every row of A has exactly one partner in B
            # and in C, and query_execute()'s planner would not run the joins of the grouped statement at all (join elimination).
            # The configuration is about the join: the timed statements run with the rule OFF; what the rule makes of the grouped statement is
            # reported beside it (`with_join_elimination`), never as `value`.
            os.environ["MDB_JOIN_ELIMINATION"] = "0"
            for name, sql, steps in (("grouped", GROUPED, args.steps), ("joined", JOINED, max(2, args.steps // 2))):
                def step(sql=sql):
                    r = db.query_device(sql, copy=False)
                    step.rows = r[3]
                    return r[4]
                dt, joined = timed(step, steps, args.warmup, db)
                res[name] = {"ms_per_step": dt / steps * 1e3, "step_ms": dict(last_dist), "value": joined / (dt / steps), "joined_rows": joined, "result_rows_rank0": step.rows,
                             "steps": steps, "kernels": prof.run(step), "plan": db.last_plan()}
            if not use_dist:
                os.environ["MDB_JOIN_ELIMINATION"] = "1"
                before = db.joins_eliminated()

                def estep():
                    r = db.query_device(GROUPED, copy=False)
                    estep.rows = r[3]
                    return r[4]
                dt, joined = timed(estep, args.steps, args.warmup, db)
                res["eliminated"] = {"ms_per_step": dt / args.steps * 1e3, "joined_rows": joined, "result_rows_rank0": estep.rows,
                                     "joins_eliminated_per_statement": (db.joins_eliminated() - before) / (max(args.warmup, 1) + 2 * args.steps),
                                     "note": "the same grouped statement with query_execute()'s default planner: B and C are not joined (the catalog's measured "
                                             "statistics say every row of A has exactly one partner in each), GROUP BY over A's primary key is the identity - "
                                             "reported for completeness, NOT the configuration's figure"}
                os.environ["MDB_JOIN_ELIMINATION"] = "0"
            g = res["grouped"]
            algo = 8 * 3 * total + 16 * g["joined_rows"]	# three key columns read once, (key, COUNT) per group written (G = joined rows here)
            algo_j = 8 * 6 * total + 8 * 6 * res["joined"]["joined_rows"]
            res["joined"]["pipeline"] = {"algorithmic_bytes": algo_j, "frac_of_peak": algo_j / (res["joined"]["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS / world}
            res["joined"]["roofline"] = roof(algo_j, res["joined"]["ms_per_step"], res["joined"]["kernels"], "config5_join",
                                             "whole statement: six 8-byte columns read once, six written per joined row")
            line.update({
                "value": g["value"], "ms_per_step": g["ms_per_step"], "steps": g["steps"],
                "config": {"workload": f"BASELINE configs[4] (SURVEY C5): A(id_a, x DOUBLE) JOIN B(id_b, y DOUBLE) JOIN C(id_c, z INT) on id_a "
                                       f"GROUP BY id_a COUNT(*), unique keys, {n} rows/table/GPU x {world} GPU, through query_execute() with "
                                       "results kept on the device",
                           "query": GROUPED, "rows_per_table_per_gpu": n, "rows_per_table_total": total, "joined_rows": g["joined_rows"],
                           "parallelism": (f"hash-partition x{world}, query_execute() in sharded mode (mdb_database_set_dist, RCCL)"
                                           + (" (forced shuffle)" if world == 1 else "")) if use_dist else "single GPU", "plan": g["plan"]},
                "pipeline": {"algorithmic_bytes": algo, "achieved_GBs": algo / (g["ms_per_step"] * 1e-3) / 1e9,
                             "frac_of_peak": algo / (g["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS / world},
                "roofline": roof(algo, g["ms_per_step"], g["kernels"], "config5", "whole statement: three key columns read once, (key, COUNT) per group written"),
                "kernels": g["kernels"],
                "join_only_form": dict(res["joined"], query=JOINED, note="x, y DOUBLE and z INT carried as payload: 6 result columns"),
                "with_join_elimination": res.get("eliminated"),
            })
            if want_cpu:
                beat("CPU baseline (rank 0)")
                line["cpu_baseline"] = reference_baseline([("A", ["id_a", "x"]), ("B", ["id_b", "y"]), ("C", ["id_c", "z"])], [("A", 2), ("B", 2), ("C", 2)],
                                                          GROUPED, 3000, "BASELINE configs[4]'s grouped statement over INT payload columns (timing only: the "
                                                          "reference's three-way join keeps the matches of C's first row alone, SURVEY D2)")
                beat("CPU check (rank 0)")
                try:    # multi-threaded CPU hash join + GROUP BY, chained over the three tables, full size
                    from oracle import cpu
                    r = db.query_device(GROUPED, copy=True)
                    cols = dict(zip(r[0], r[2]))
                    gk, gc = cols["A.id_a"].cpu().numpy(), cols["COUNT(*)"].cpu().numpy()
                    keys = [db.query_device(f"SELECT {c} FROM {t};", copy=True)[2][0].cpu().numpy() for t, c in (("A", "id_a"), ("B", "id_b"), ("C", "id_c"))]
                    t0 = time.perf_counter()
                    k1, c1, _, _ = cpu.hash_join_group_count(keys[0], None, keys[1], None, os.cpu_count() or 1)
                    k2, c2, _, j2 = cpu.hash_join_group_count(k1, None, keys[2], None, os.cpu_count() or 1)
                    sec = time.perf_counter() - t0
                    # (groups of (A, B), then those keys against C: the counts multiply - all 1 here: unique keys)
                    line["cpu_check"] = {"what": "oracle/cpu_hash.c chained over the three tables (multi-threaded), the benchmark's own tables, full size",
                                         "cores": os.cpu_count() or 1, "seconds": sec, "value": j2 / sec, "unit": "joined rows/s",
                                         "gpu_result_identical": bool(bool((c1 == 1).all()) and np.array_equal(gk, k2) and np.array_equal(gc, c2))}
                except Exception as e:  # pragma: no cover
                    line["cpu_check"] = {"error": str(e)}
            else:
                line["cpu_baseline"] = None
    if not want_cpu:
        line["cpu_baseline_note"] = "--no-cpu-baseline" if args.no_cpu_baseline else "rank 0 at N = 1 only"
    if use_dist:    # (make_dist() has checked that the library's own communicator saw every rank)
        line["config"]["rccl_ranks_seen"] = "test transport" if host_wire else world
        if host_wire:
            line["config"]["transport"] = "host memory through a gloo process group, all ranks on GPU 0 (--transport test)"
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
