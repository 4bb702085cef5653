#!/usr/bin/env python3
"""bench.py - joined rows/sec of the north-star query on 1..N MI355X (contract: see DESIGN.md 6).

    SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a;

A "step" = one pass of the whole device pipeline (hash, partition both tables, per-leaf LDS hash
build/probe, ordered group emission) over synthetic INT64 tables that are already resident in
HBM when the timed region starts.  N = 1: 10^8 rows per table on one GPU (BASELINE.json
configs[2]).  N > 1 (launched by torch.distributed.run, one rank per GPU): every rank holds 10^8
rows of each table (weak scaling), keys are hash-partitioned by destination GPU, exchanged with
one RCCL all-to-all per table over xGMI, then joined locally; no other collective is on the path.

The JSON line carries `roofline` for the dominant kernel (live HIP-event timing through the
library's per-kernel profiler) and `cpu_baseline` (the real reference executor if
oracle/_ref/libmidori_ref.so travelled with the snapshot, else the repo's own C restatement).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)

# names of the profiler's kernels in the rocprofv3 summary kept under profiles/ (PMC traffic per launch)
ROCPROF_NAMES = {
    "leaf_join_group_count": ["k_leaf_group_count<true, true, false, true>", "k_leaf_group_count<true, false, false, true>",
                              "k_leaf_group_count<true, true, false, false>", "k_leaf_group_count<true, false, false, false>"],
    "leaf_group_count": ["k_leaf_group_count<false, false, false, true>", "k_leaf_group_count<false, false, false, false>"],
    "part_hist_l0": ["k_part_hist<true, false>"],
    # level 0 / level 1 of the histogram-free partition: left table (8-byte words), right table (4-byte words in the narrow form)
    "part_scatter_l0": ["k_part_scatter<true, false, false, true, false, false, false>", "k_part_scatter<true, false, false, true, false, true, false>",
                        "k_part_scatter<true, true, false, true, false, false, false>"],
    "part_scatter_l1": ["k_part_scatter<false, false, false, true, false, false, false>", "k_part_scatter<false, false, false, true, false, true, false>",
                        "k_part_scatter<false, true, false, true, false, false, false>"],
    "order_leaf": ["k_order_leaf"],
    "gather64": ["k_gather64"],
}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r01/rocprof_summary.json:
    FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE), or None.  Counters cannot be collected from inside
    this process; the summary comes from `bash profiles/collect.sh` on the same workload (10^8 rows, variant D)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01", "rocprof_summary.json")) as f:
            ks = json.load(f)["kernels"]
        vals = [ks[n]["hbm_read_bytes"] + ks[n]["hbm_write_bytes"] for n in ROCPROF_NAMES.get(kernel, []) if n in ks]
        return sum(vals) / len(vals) if vals else None
    except Exception:
        return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=100_000_000, help="rows per table per GPU")
    ap.add_argument("--variant", choices=["U", "D"], default="D",
                    help="U: both key columns are permutations (1:1); D: B keys = perm mod N/16 (1:16 duplicates)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true", help="check the result against the CPU oracle (rows <= 2e7)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary variant-U measurement (profiling runs)")
    ap.add_argument("--wire64", action="store_true", help="multi-GPU exchange: always ship 8-byte keys (default: 4-byte keys when the "
                    "column statistics allow it)")
    ap.add_argument("--chunks", type=int, default=None, help="pieces per table in the multi-GPU exchange (default 1)")
    ap.add_argument("--force-shuffle", action="store_true",
                    help="run the multi-GPU pipeline (partition by destination + RCCL all-to-all + local join) even with one rank")
    return ap.parse_args()


def cpu_baseline():
    """Time the CPU path on this box's host cores on a bounded sample of the same query.

    Preferred: the REAL reference executor (oracle/_ref, built in the authoring container from
    /root/reference and shipped with the snapshot) on 5000 x 5000 rows (about 15 s) through its own
    ast -> semantic -> optimiser -> executor pipeline.  Fallback: oracle/cpu_naive.c ("port").
    """
    n = 5000
    try:
        from oracle import ref as refmod
        if refmod.available():
            times = {}
            for m in (n // 2, n):		# two sizes: the second shows the quadratic growth the extrapolation uses
                db = refmod.RefDB()
                a = np.arange(m, dtype=np.int64)
                rng = np.random.default_rng(42)
                db.create_int_table("A", ["id_a"])
                db.create_int_table("B", ["id_b"])
                db.bulk_insert("A", [rng.permutation(a)])
                db.bulk_insert("B", [rng.permutation(a)])
                t0 = time.perf_counter()
                cols, rows = db.query("SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a;")
                times[m] = time.perf_counter() - t0
                db.close()
            dt = times[n]
            pairs_s = n * n / dt
            return {"value": n / dt, "unit": "joined rows/s", "cores": 1, "kind": "reference",
                    "sample": f"north-star query, {n}x{n} unique keys, real reference executor via oracle/_ref ({dt:.2f} s; "
                              f"{n // 2}x{n // 2}: {times[n // 2]:.2f} s, i.e. O(nA*nB) nested loop at {pairs_s:.3g} row pairs/s; "
                              f"10^8 x 10^8 rows would take {1e16 / pairs_s / 3.15e7:.0f} years)"}
    except Exception as e:  # pragma: no cover - diagnostic path
        sys.stderr.write(f"[bench] reference baseline unavailable: {e}\n")
    from oracle import cpu
    rng = np.random.default_rng(42)
    a = rng.permutation(np.arange(n, dtype=np.int64))
    b = rng.permutation(np.arange(n, dtype=np.int64))
    t0 = time.perf_counter()
    cpu.naive_join_group_count(a, None, b, None)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "joined rows/s", "cores": 1, "kind": "port",
            "sample": f"north-star query, {n}x{n} unique keys, oracle/cpu_naive.c nested loop ({dt:.2f} s)"}


def cpu_hash_yardstick(a_dev, b_dev, gpu_result=None):
    """the repo's multi-threaded CPU hash join + aggregate (oracle/cpu_hash.c, pinned to the reference's
    vectors) on all host cores, on the SAME tables the GPU just processed - at the full configuration size when the host
    has the cores for it (about 2 s on the GPU box), otherwise on the first 2*10^7 rows of each.  At full size its result
    is also the parity check of the benchmark run itself: keys, counts and order of the GPU's result are compared."""
    from oracle import cpu
    cores = os.cpu_count() or 1
    rows = a_dev.numel()
    full = cores >= 32 and rows <= 200_000_000
    n = rows if full else min(rows, 20_000_000)
    a = a_dev[:n].cpu().numpy()
    b = b_dev[:n].cpu().numpy()
    t0 = time.perf_counter()
    ek, ec, _, j = cpu.hash_join_group_count(a, None, b, None, cores)
    dt = time.perf_counter() - t0
    out = {"value": j / dt, "unit": "joined rows/s", "cores": cores, "rows_per_table": n, "seconds": dt,
           "sample": "the benchmark's own tables, full size" if full else "first rows of the benchmark's tables"}
    if full and gpu_result is not None:
        k, c, jj = gpu_result
        out["gpu_result_identical"] = bool(jj == j and k.numel() == len(ek) and np.array_equal(k.cpu().numpy(), ek)
                                           and np.array_equal(c.cpu().numpy(), ec))
    return out


def main():
    args = parse_args()
    # stdout carries exactly ONE line (the JSON): everything else that native libraries print there
    # (e.g. RCCL's version banner) is routed to stderr at the file-descriptor level
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_shuffle
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from midoridb_amd.dev import DeviceCtx
    from midoridb_amd import shuffle

    dev = DeviceCtx(local_rank)
    n = args.rows
    total_rows = n * world
    mod_b = total_rows // 16 if args.variant == "D" else 0
    # rank r holds rows [r*n, (r+1)*n) of the global tables (pre-sharded round-robin is equivalent for a permutation)
    a = dev.gen_keys(n, rank * n, total_rows, 42, 0)
    b = dev.gen_keys(n, rank * n, total_rows, 43, mod_b)
    cap = int(n * 1.3) + 4096 if use_dist else n
    out = (torch.empty(cap, dtype=torch.int64, device=dev.device), torch.empty(cap, dtype=torch.int64, device=dev.device),
           torch.empty(cap, dtype=torch.int32, device=dev.device))
    wire32 = False
    if use_dist and not args.wire64:
        # column statistics (computed once per table, outside the timed region, like a catalog would keep them):
        # when every key of both columns fits 32 bits on every rank the exchange ships 4-byte keys
        fits = 1.0
        for col in (a, b):
            lo, hi = dev.key_range(col)
            if lo > hi or lo < -(1 << 31) or hi >= (1 << 31):
                fits = 0.0
        t = torch.tensor([fits], dtype=torch.float64, device=dev.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        wire32 = bool(t.item() > 0.5)
    pipeline = shuffle.DistributedJoinGroupCount(dev, world, rank, n, chunks=args.chunks, wire32=wire32) if use_dist else None

    def step():
        if pipeline is None:
            k, c, f, j = dev.join_group_count(a, None, b, None, out=out)
            return k.numel(), j
        return pipeline.run(a, b, out)

    for _ in range(max(args.warmup, 1) if world == 1 else args.warmup):
        g, j = step()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g, j = step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev.device)
    jsum = torch.tensor([float(j)], dtype=torch.float64, device=dev.device)
    gsum = torch.tensor([float(g)], dtype=torch.float64, device=dev.device)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(jsum, op=dist.ReduceOp.SUM)
        dist.all_reduce(gsum, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    joined_total = int(jsum.item())
    groups_total = int(gsum.item())
    ms_per_step = dt / args.steps * 1e3
    value = joined_total / (dt / args.steps)

    # ---- per-kernel profile (outside the timed region): live HIP events around every launch
    dev.prof_enable(True)
    dev.prof_reset()
    prof_steps = 3
    for _ in range(prof_steps):
        step()
    prof = dev.prof_read()
    dev.prof_enable(False)
    # practical HBM ceiling of this box: device-to-device copy of one key column (read + write)
    copy_gbs = None
    try:
        src = a
        dst = torch.empty_like(a)
        for _ in range(2):
            dst.copy_(src)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * src.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del dst
    except Exception:
        pass

    if rank == 0:
        narrow = dev.last_join_narrow()
        kern = {k: {"launches_per_step": v[0] / prof_steps, "ms_per_step": v[1] / prof_steps} for k, v in prof.items()}
        for k, d in kern.items():   # per-kernel achieved rate on its algorithmic bytes
            if d["ms_per_step"] > 0:
                d["algorithmic_GBs"] = (shuffle.algorithmic_bytes(k, n, groups_total / max(world, 1), narrow) * d["launches_per_step"]
                                        / (d["ms_per_step"] * 1e-3) / 1e9)
        # dominant kernel = the level-0/1 scatter; algorithmic bytes of one launch = every key it moves,
        # read once (8 B hashed key [+4 B row id]) and written once
        dom_name = max(kern, key=lambda k: kern[k]["ms_per_step"]) if kern else None
        roof = None
        if dom_name:
            d = kern[dom_name]
            launches = max(d["launches_per_step"], 1e-9)
            avg_ms = d["ms_per_step"] / launches
            bytes_per_launch = shuffle.algorithmic_bytes(dom_name, n, groups_total / max(world, 1), narrow)
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            roof = {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS,
                    "traffic": pmc_traffic(dom_name) if (n == 100_000_000 and args.variant == "D" and world == 1) else None,
                    "d2d_copy_GBs": copy_gbs,
                    "frac_of_d2d_copy": (achieved / copy_gbs) if copy_gbs else None,
                    "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_per_launch}
        # whole-pipeline view: the bytes any correct algorithm must move once
        algo_bytes = 8 * 2 * total_rows + 16 * groups_total
        line = {
            "metric": "joined rows/sec, 2x10^8-row INT64 INNER JOIN+GROUP BY, 1/2/4/8 MI355X",
            "value": value, "unit": "joined rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64", "data": "synthetic",
            "config": {"workload": f"A JOIN B ON id_a=id_b GROUP BY id_a COUNT(*), {n} rows/table/GPU, variant {args.variant} "
                                   f"({'B keys 16x duplicated' if args.variant == 'D' else 'unique keys both sides'})",
                       "key_form": "narrow (keys within one 2^32-wide window, verified on the device: 32-bit hashes)" if narrow else "wide (64-bit hashes)",
                       "rows_per_table_per_gpu": n, "joined_rows": joined_total, "groups": groups_total,
                       "order": "reference first-occurrence order" if not use_dist else "per rank, first occurrence in the received stream",
                       "parallelism": f"hash-partition x{world}" + (" (forced shuffle)" if args.force_shuffle and world == 1 else "")
                                      + ((", 4-byte keys on the wire" if wire32 else ", 8-byte keys on the wire") if use_dist else "")},
            "roofline": roof,
            "pipeline": {"algorithmic_bytes": algo_bytes, "achieved_GBs": algo_bytes / (dt / args.steps) / 1e9,
                         "frac_of_peak": algo_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS},
            "kernels": kern,
        }
        if world == 1 and not use_dist and args.variant == "D" and not args.no_secondary:
            # the other synthetic variant of SURVEY 8d C3 (unique keys on both sides: G = n groups), same pipeline
            try:
                b_u = dev.gen_keys(n, 0, n, 43, 0)
                for _ in range(2):
                    dev.join_group_count(a, None, b_u, None, out=out)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                reps = max(3, args.steps // 2)
                for _ in range(reps):
                    ku, cu, fu, ju = dev.join_group_count(a, None, b_u, None, out=out)
                torch.cuda.synchronize()
                dtu = (time.perf_counter() - t1) / reps
                line["variant_U"] = {"workload": f"unique keys both sides, {n} rows/table", "joined_rows": ju, "groups": int(ku.numel()),
                                     "ms_per_step": dtu * 1e3, "value": ju / dtu}
                del b_u
            except Exception as e:  # pragma: no cover
                line["variant_U"] = {"error": str(e)}
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            try:
                gpu_result = None
                if pipeline is None:
                    k, c, f, jj = dev.join_group_count(a, None, b, None, out=out)
                    gpu_result = (k, c, jj)
                line["cpu_hash"] = cpu_hash_yardstick(a, b, gpu_result)
            except Exception as e:  # pragma: no cover
                line["cpu_hash"] = {"error": str(e)}
        if args.verify and n <= 20_000_000 and world == 1:
            from oracle import cpu, np_oracle as orc
            ek, ec, ef, ej = cpu.hash_join_group_count(orc.gen_keys(n, 0, n, 42, 0), None, orc.gen_keys(n, 0, n, 43, mod_b),
                                                       None, os.cpu_count() or 1)
            if pipeline is None:
                k, c, f, jj = dev.join_group_count(a, None, b, None, out=out)
                ok = (jj == ej and np.array_equal(k.cpu().numpy(), ek) and np.array_equal(c.cpu().numpy(), ec))
            else:   # shuffled pipeline: same groups, order is not the reference's
                _, jj = pipeline.run(a, b, out)
                k, c, _ = pipeline.last
                o1, o2 = np.argsort(k.cpu().numpy(), kind="stable"), np.argsort(ek, kind="stable")
                ok = (jj == ej and np.array_equal(k.cpu().numpy()[o1], ek[o2]) and np.array_equal(c.cpu().numpy()[o1], ec[o2]))
            line["verified_vs_oracle"] = bool(ok)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    os.close(json_fd)


if __name__ == "__main__":
    main()
