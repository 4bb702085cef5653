#!/usr/bin/env python3
"""bench.py - joined rows/sec of the north-star query on 1..N MI355X (contract: see DESIGN.md 6).

    SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a;

A "step" = one pass of the whole device pipeline (hash, partition both tables, per-leaf LDS
build/probe, ordered group emission) over synthetic INT64 tables that are already resident in
HBM when the timed region starts.  N = 1: 10^8 rows per table on one GPU (BASELINE.json
configs[2]).  N > 1, one rank per GPU: keys are hash-partitioned by destination GPU, exchanged with
one RCCL all-to-all per table over xGMI, then joined locally; no other collective is on the path.
The primary line is WEAK scaling (every rank holds --rows rows of each table); the same run also
measures the STRONG form the metric's wording suggests (2 x 10^8 rows in total, split over the
ranks) and reports it under "strong_scaling".

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own ranks
(`python -m torch.distributed.run` as a child process, before this process touches a GPU), relays
rank 0's JSON line and exits with the children's code; under torch.distributed.run it is a rank.

The JSON line carries `roofline` for the kernel that takes the most time per step (live HIP-event
timing through the library's per-kernel profiler; kernel names = template instances, so the two
tables' partition launches are not lumped together), `cpu_baseline` (the real reference executor
if oracle/_ref/libmidori_ref.so travelled with the snapshot, else the repo's own C restatement),
and, at N = 1: the wide (64-bit hash) form, the other synthetic variant, cold-start figures and the
same query end to end through query_execute().
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = "joined rows/sec, 2x10^8-row INT64 INNER JOIN+GROUP BY, 1/2/4/8 MI355X"
NORTH = "SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a;"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)

PROFILE_ROUNDS = ("r06", "r05", "r04", "r03", "r02", "r01")


def _pmc_summary(suffix):
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", rnd, f"rocprof_summary{suffix}.json")
        try:
            with open(path) as f:
                return json.load(f)["kernels"], f"profiles/{rnd}/rocprof_summary{suffix}.json"
        except Exception:
            continue
    return None, None


def pmc_traffic(rocprof_names, variant="D"):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/rNN/rocprof_summary*.json: FETCH_SIZE x2 per
    the gfx950 correction + WRITE_SIZE), newest round first, or None.  `rocprof_names` = the names rocprofv3 lists the kernel's
    template instances under; they come from the library itself (mdb_dev_prof_symbols: the symbols of what was launched under the
    profiler name), so nothing here spells a mangled name; several instances are weighted by their launch counts in the profiled
    run.  Counters cannot be collected from inside this process; the summary comes from `bash profiles/collect.sh` on the same
    workload (10^8 rows, the same variant: suffix "" = D, "_U", "_S", "_wide", "_shuffle")."""
    suffix = "" if variant == "D" else f"_{variant}"
    ks, src = _pmc_summary(suffix)
    if ks is None:
        return None
    num = den = 0.0
    for n in rocprof_names:
        d = ks.get(n)
        if d and "hbm_read_bytes" in d and "hbm_write_bytes" in d:
            w = float(d.get("calls", 1) or 1)
            num += w * (d["hbm_read_bytes"] + d["hbm_write_bytes"])
            den += w
    return {"bytes": num / den, "source": src} if den else None


def pmc_step_traffic(kern, variant, algo_bytes):
    """HBM bytes ONE step moves by the PMC counters: per profiler name, bytes per launch (committed summary of the same workload) x
    the launches per step measured live in this run; over the whole-query algorithmic bytes of SURVEY 8(d).  None when a summary for
    this variant is not committed or does not cover a kernel that ran."""
    total, src, missing = 0.0, None, []
    for name, d in kern.items():
        tr = pmc_traffic(d.get("rocprof_names", []), variant)
        if tr is None:
            missing.append(name)
            continue
        total += tr["bytes"] * d["launches_per_step"]
        src = tr["source"]
    if src is None:
        return None
    return {"pmc_bytes_per_step": total, "algorithmic_bytes": algo_bytes, "traffic_over_algorithmic": total / algo_bytes if algo_bytes else None,
            "source": src, "kernels_without_counters": missing}


def scatter_word_bytes(instances):
    """(bytes read, bytes written) per row by the k_part_scatter instance(s) that ran, from the traits in their names - pf_key*: 8-byte
    keys in, pf_word*: words in; _w32: 4-byte words; _out16: 2-byte words out; _rid: a 4-byte row id beside the word.  The instance
    that RAN decides, not the profiler name it ran under (forced shuffle runs both tables under part_scatter_l0_w32)."""
    ins, outs = [], []
    for name in instances:
        i = name.find("k_part_scatter<")
        if i < 0:
            continue
        t = name[i + len("k_part_scatter<"):].split(">")[0]
        rid = 4 if "_rid" in t else 0
        w32, out16 = "_w32" in t, "_out16" in t
        if t.startswith("pf_key"):
            ins.append(8)
        else:
            ins.append((4 if w32 else 8) + rid)
        outs.append((2 if out16 else 4 if w32 else 8) + rid)
    if not ins:
        return None
    return sum(ins) / len(ins), sum(outs) / len(outs)


FIRST_LEVEL = ("part_scatter_l0", "part_scatter_l0_w32", "part_scatter_l0_rid", "part_scatter_l0_pruned", "part_hist_l0",
               "shard_scatter_wide_l", "shard_scatter_wide_r", "part_scatter_wide12_l", "part_scatter_wide12_r")
WRITES_RESULT = ("order_leaf", "order_leaf_sparse", "shard_leaf", "shard_leaf_wide")


def survey_bytes(kernel, n, groups):
    """SURVEY 8(d)'s algorithmic bytes that ONE launch of `kernel` consumes: the whole query owes 8 (nA + nB) bytes of keys read and
    16 G bytes of result written, once, whatever the passes in between.  A first-level pass is where a table's 8 n key bytes are
    read; the kernel that writes the groups owns the 16 G; every pass in between (second levels, leaves that write records, the
    ordering sort) moves bytes SURVEY 8(d) does not count - None: such a kernel has only its own I/O to be priced against."""
    if kernel in FIRST_LEVEL:
        return 8.0 * n
    if kernel in WRITES_RESULT:
        return 16.0 * groups
    return None


def algorithmic_bytes(kernel, n, groups, narrow, pruned=False, levels=2, instances=(), launches=1.0, left_kept=None, keys_alias=False):
    """The kernel's OWN I/O per launch: what the launch must read once and write once given the words this design
    moves - `roofline.frac_kernel_io`, next to the SURVEY 8(d) figure of survey_bytes().  Kernel names are template instances, one
    table each; the scatter kernels' word sizes come from the instance that ran (scatter_word_bytes).
    pruned: min-max pruning ran - the left table's first level wrote only the rows inside the right table's key range, and the
    kernels after it see those rows only (one left row per group in both benchmark variants: left_kept = G)."""
    key, rid, h32, g = 8 * n, 4 * n, 4 * n, groups
    kept = left_kept if left_kept is not None else (g if pruned else n)
    wb = scatter_word_bytes(instances)
    if wb is not None and kernel.startswith("part_scatter_l"):
        bin_, bout = wb
        if kernel == "part_scatter_l0_pruned":
            return float(bin_ * n + bout * kept)
        if kernel.startswith("part_scatter_l0"):
            if launches >= 1.5:     # both tables under one name (the sharded operator's sender): the right table whole, the left one pruned
                return float((bin_ * n + bout * n + bin_ * n + bout * kept) / 2)
            return float(bin_ * n + bout * n)
        rows = kept if (pruned and kernel in ("part_scatter_l1", "part_scatter_l1_semi")) else n
        if kernel == "part_scatter_l1_semi":
            return float(bin_ * rows + bout * g)
        return float((bin_ + bout) * rows)
    if levels == 1:         # one partition level: the right table travels as 2-byte words (the hash bits below the first level's digit)
        t = {"leaf_join_wide": (8 * g if pruned else key) + 2 * n + 8 * g, "leaf_join_wide4": (8 * g if pruned else key) + 2 * n + 8 * g}
        if kernel in t:
            return float(t[kernel])
    if pruned:
        t = {"leaf_join_direct": 8 * g + h32 + 8 * g,
             "leaf_join_wide": 8 * g + h32 + 8 * g, "leaf_join_wide4": 8 * g + h32 + 8 * g}
        if kernel in t:
            return float(t[kernel])
    table = {
        "part_scatter_l0": key + key,               # read 8-byte keys, write 8-byte words (hash | row id in the narrow form)
        "part_scatter_l0_w32": key + h32,           # narrow right side: read keys, write 4-byte hashes
        "part_scatter_l0_rid": key + key + rid,     # wide left side: hash + row id out
        "part_scatter_l1": key + key,
        "leaf_bitmap": h32,                         # the right table's partitioned words in, a bitmap of 2^k / 2^c bits out
        "part_scatter_l1_w32": h32 + h32,
        "part_scatter_l1_rid": 2 * (key + rid),
        "part_hist_l0": key,
        "shard_scatter_wide_l": key + 2 * kept, "shard_scatter_wide_r": key + 2 * n,
        # the ordered form's one 4096-digit pass per table: 2-byte words (right), 4-byte row words + one header per ~8-word run (left)
        "part_scatter_wide12_r": key + 2 * n, "part_scatter_wide12_l": key + 4.5 * kept,
        "leaf_join_wide12": 2 * n + 4.5 * kept + 8 * g,     # both tables' words in, one record per group out (4 bytes when every COUNT fits)
        # ... the bit-per-row form: the same words in, one bit per left row touched (cleared where the row is no group's first), no record out
        "leaf_join_wide12_bits": 2 * n + 4.5 * kept + n / 8,
        "dense_count": n / 8,                       # the bits, counted per block of rows
        # the bits and the left key column in, (first row, key, COUNT) per group out; MDB_KEYS_MAY_ALIAS and every left row a group: the key
        # column is neither read nor written (the caller reads the left column itself), and this benchmark asks for no first-row column -
        # COUNT per group out
        "dense_expand": n / 8 + 8 * g if keys_alias else n / 8 + key + 20 * g,
        "dense_patch": 16,                          # the exceptions' COUNTs (a handful)
        "expand_keys": 16 * g + key,                # (key, COUNT) per group in, every key COUNT times out
        "shard_leaf_wide": 2 * (kept + n) + 16 * g, "shard_leaf": 4 * (kept + n) + 16 * g,
        "leaf_join_group_count": (key + h32 if narrow else key + rid + key) + 8 * g,    # both partitioned tables in, one record per group out
        "leaf_join_direct": key + h32 + 8 * g,
        "leaf_join_wide": key + h32 + 8 * g,         # one partition level: the first level's words in, one record per group out
        "leaf_join_wide4": key + h32 + 8 * g,        # ... with 4-byte table entries (two workgroups per CU)
        "leaf_group_wide": key + 8 * g,
        "order_leaf_sparse": 8 * g + 20 * g,        # keyed records in, (first row, key, COUNT) out
        "leaf_group_count": (key if narrow else key + rid) + 8 * g,
        "sort_hist_l0": 8 * g,
        "sort_scatter_l0": 16 * g,
        "sort_scatter_l1": 16 * g,
        "order_leaf": 8 * g + 8 * g + 16 * g,       # records + gathered keys in, (key, COUNT) out
        "gather64": 4 * g + 8 * g + 8 * g,
        # the reduce over the right pass's per-tile (min, max) pairs: 8 bytes per tile of 4096 / 8192 keys, not the table
        "part_minmax": 8 * (n / 4096 + 8), "key_sample": 2 * 4096 * 8, "shard_regions": 4 * 512 * 8 * 2,
    }
    small = ("scan_", "part_build_tiles", "part_region_tiles", "part_seg0", "part_children", "shard_tiles", "order_ranges", "key_range")
    if kernel not in table and kernel.startswith(small):
        return 0.0      # descriptor / scan kernels over a few thousand words: no table bytes to price them on
    # (a kernel this table does not know is NOT priced - it used to be priced as "one key column", which gave dense_count, 12.5 MB of
    # bits, a fraction of 8.55 in round 5's tables; profiles/make_tables.py refuses an unpriced kernel above 0.02 ms and a fraction > 1)
    return float(table.get(kernel, 0.0))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=None, help="rows per table per GPU (weak scaling; default 10^8, 10^7 for --config 2); 125000000 = BASELINE "
                    "configs[3] on 8 GPUs")
    ap.add_argument("--variant", choices=["U", "D", "S"], default="D",
                    help="U: both key columns are permutations (1:1); D: B keys = perm mod N/16 (1:16 duplicates, all in the lowest sixteenth "
                         "of A's key range); S: the same duplication SPREAD over A's whole range, B keys = 16 * (perm mod N/16) - no range "
                         "for min-max pruning to use, no small key window")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="which form is the primary line at N > 1 (the other one is reported beside it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true", help="check the result against the CPU oracle (rows <= 2e7)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip everything but the primary measurement and the per-kernel profile (profiling runs)")
    ap.add_argument("--wire64", action="store_true", help="multi-GPU exchange: always ship 8-byte keys (default: 4-byte keys when the "
                    "column statistics allow it)")
    ap.add_argument("--config", type=int, choices=[2, 3, 4, 5], default=3,
                    help="which BASELINE.json configuration (SURVEY 8d numbering): 3 = the north-star query (default, the metric's own "
                         "configuration), 2 = configs[1]: two-table join with payload at 10^7 rows through query_execute(), 4 = configs[3]: SELECT * join over key columns (10^9 rows over 8 GPUs: --rows 125000000), "
                         "5 = configs[4]: three-way join + GROUP BY with DOUBLE payload through query_execute() (bench_configs.py)")
    ap.add_argument("--unordered", action="store_true", help="N = 1: time the operator without MDB_ORDER_FIRST (groups in unspecified order) - "
                    "evidence runs only; the default line keeps the reference's first-occurrence order")
    ap.add_argument("--reference-order", action="store_true", help="--config 4 at N = 1: time mdb_dev_join_pairs + the key gather (joined rows in "
                    "the reference's left-major order) instead of mdb_dev_join_keys (unspecified order, like the sharded form)")
    ap.add_argument("--transport", choices=["rccl", "test"], default="rccl",
                    help="N > 1: `rccl` = one rank per GPU, RCCL over xGMI (the measurement); `test` = N ranks on GPU 0, the blocks carried "
                         "through host memory by a gloo process group (DistCtx.over_host_group) - runs the N > 1 code of this file on a "
                         "one-GPU box; its timings say nothing about xGMI")
    ap.add_argument("--peer-timeout", type=float, default=600.0,
                    help="N > 1: seconds a rank waits in one phase (a collective, a step) before it gives up and EXITS non-zero - a rank "
                         "that lost a peer must not hang the job")
    ap.add_argument("--force-shuffle", action="store_true",
                    help="run the multi-GPU pipeline (partition by destination + RCCL all-to-all + local join) even with one rank")
    args = ap.parse_args()
    if args.config == 3 and not args.rows:
        args.rows = 100_000_000
    return args


def launch_ranks(args):
    """N > 1 without a launcher: start the ranks as CHILD processes (never re-exec: a process that has touched the GPU
    must not be replaced, and this one has not touched it - device_count() does not initialise HIP on this image),
    relay rank 0's line, return the children's exit code."""
    visible = torch.cuda.device_count()
    if visible < (1 if args.transport == "test" else args.gpus):
        line = {"metric": METRIC, "value": None, "unit": "joined rows/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                "skipped": f"--gpus {args.gpus} needs {args.gpus} visible GPUs, this box has {visible}"}
        print(json.dumps(line), flush=True)
        sys.stderr.write(f"[bench] {line['skipped']}\n")
        return 0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif p.returncode == 0:
        sys.stderr.write("[bench] the ranks exited without a result line\n")
        return 1
    return p.returncode


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
                cols, rows = db.query(NORTH)
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


def cpu_naive_sizes():
    """the repo's restatement of the reference algorithm (oracle/cpu_naive.c: nested-loop join with early
    materialisation, quadratic GROUP BY), one thread, at N = 1000 / 2000 / 4000 rows per table, with the quadratic
    extrapolation to the configuration's size."""
    from oracle import cpu
    out = {"cores": 1, "kind": "port", "sizes": {}}
    rng = np.random.default_rng(42)
    pairs_s = 0.0
    for n in (1000, 2000, 4000):
        a = rng.permutation(np.arange(n, dtype=np.int64))
        b = rng.permutation(np.arange(n, dtype=np.int64))
        t0 = time.perf_counter()
        cpu.naive_join_group_count(a, None, b, None)
        dt = time.perf_counter() - t0
        pairs_s = n * n / dt
        out["sizes"][str(n)] = {"seconds": dt, "row_pairs_per_s": pairs_s, "joined_rows_per_s": n / dt}
    out["extrapolated_seconds_at_1e8_x_1e8"] = 1e16 / pairs_s
    return out


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


def end_to_end(n, mod_b, a_dev, b_dev):
    """The same query through the drop-in C API (query_execute -> plan -> device pipeline -> D2H of the result into
    page-locked host columns), reported beside `value`, never as it: (1) tables resident in the device
    mirror, (2) tables that arrive as host columns, so the first SELECT also uploads 2 x 8n bytes over PCIe."""
    from midoridb_amd.query import DB
    out = {}
    with DB() as db:
        db.execute("CREATE TABLE A (id_a INT);")
        db.execute("CREATE TABLE B (id_b INT);")
        db.generate("A", n, 42, [0])
        db.generate("B", n, 43, [mod_b])
        walls, execs, rows = [], [], 0
        for _ in range(4):
            r = db.query(NORTH)
            walls.append(db.last_call_ms)		# the query_execute() call itself
            execs.append(r.exec_ms)
            rows, joined = r.nrows, r.joined_rows
        db.results_on_device(True)       # the same call without the device-to-host copy of the result (mdb_database_results_on_device)
        dwalls = []
        for _ in range(4):
            db.query_device(NORTH, copy=False)
            dwalls.append(db.last_call_ms)
        db.groups_any_order(True)        # ... and without the reference's first-occurrence group order (mdb_database_groups_any_order)
        awalls = []
        for _ in range(4):
            db.query_device(NORTH, copy=False)
            awalls.append(db.last_call_ms)
        db.groups_any_order(False)
        db.results_on_device(False)
        out["device_resident_any_group_order"] = {"wall_ms": min(awalls[1:]), "value": joined / (min(awalls[1:]) * 1e-3),
                                                  "includes": "as device_resident_tables_and_result, groups in unspecified order"}
        out["device_resident_tables_and_result"] = {"wall_ms": min(dwalls[1:]), "value": joined / (min(dwalls[1:]) * 1e-3),
                                                    "includes": "wall time of query_execute(): SQL parse + plan + device pipeline; the result "
                                                                "columns stay in HBM (query_column_data_device), fetched on first cursor use"}
        out["device_resident_tables"] = {"wall_ms": min(walls[1:]), "executor_ms": min(execs[1:]), "first_call_wall_ms": walls[0],
                                         "result_rows": rows, "joined_rows": joined, "value": joined / (min(walls[1:]) * 1e-3),
                                         "includes": "wall time of query_execute(): SQL parse + plan + device pipeline + D2H of the result columns"}
    ha, hb = a_dev.cpu().numpy(), b_dev.cpu().numpy()
    with DB() as db:
        db.execute("CREATE TABLE A (id_a INT);")
        db.execute("CREATE TABLE B (id_b INT);")
        # the device context and its scratch arena exist before the clock starts, as in a server that has run a query before: a fresh
        # context's first hipMalloc of the 9.1 GB arena this statement asks for (profiles/micro/arena_need_db.py) takes 3 - 600 ms
        # depending on the box and on what the process has freed before, and would drown what is measured here
        from midoridb_amd.dev import _bind as _bind_dev
        _bind_dev(db.lib)
        t0 = time.perf_counter()
        db.lib.mdb_dev_reserve(db.device_handle(), 10 << 30)
        reserve_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        db.append_columns("A", [ha])
        db.append_columns("B", [hb])
        ingest_ms = (time.perf_counter() - t0) * 1e3
        r = db.query(NORTH)
        first_ms = db.last_call_ms
        r = db.query(NORTH)
        again_ms = db.last_call_ms
        out["host_resident_tables"] = {"context_and_arena_ms": reserve_ms, "bulk_ingest_ms": ingest_ms, "first_select_wall_ms": first_ms, "second_select_wall_ms": again_ms,
                                       "h2d_bytes": 16 * n, "value_first_select": r.joined_rows / (first_ms * 1e-3),
                                       "includes": "bulk_ingest_ms: mdb_table_append_columns of both tables from host arrays - copied into the host store in chunks by "
                                                   "several threads while the previous chunk goes up into the device mirror (pageable H2D); the first SELECT "
                                                   "then uploads nothing (round 4: 260 ms + a 58-85 ms first SELECT)"}
    return out


class Watchdog:
    """N > 1: a rank that lost a peer sits in a collective for ever (RCCL kernels do not time out).  A thread watches the main thread's
    heartbeat - one per phase - and, when a phase outlasts the timeout, says where and EXITS the process non-zero (os._exit: no re-exec,
    no clean-up that could block on the device), so the launcher sees a failed rank and ends the others."""

    def __init__(self, timeout_s, rank):
        import threading
        self.timeout, self.rank, self.t, self.what, self.on = float(timeout_s), rank, time.monotonic(), "start-up", True
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def beat(self, what):
        self.t, self.what = time.monotonic(), what

    def stop(self):
        self.on = False

    def _run(self):
        while self.on:
            time.sleep(min(1.0, self.timeout / 4))
            if self.on and time.monotonic() - self.t > self.timeout:
                sys.stderr.write(f"[bench] rank {self.rank}: no progress for {self.timeout:.0f} s in phase '{self.what}' - a peer is gone "
                                 "or a collective hangs; exiting non-zero\n")
                sys.stderr.flush()
                os._exit(3)


def dist_of(ms):
    """min / median / p90 / max of per-step times (ms)"""
    v = sorted(ms)
    n = len(v)
    return {"steps": n, "min": v[0], "median": v[n // 2], "p90": v[min(n - 1, (9 * n) // 10)], "max": v[-1], "mean": sum(v) / n}


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    # stdout carries exactly ONE line (the JSON): everything else that native libraries print there
    # (e.g. RCCL's version banner) is routed to stderr at the file-descriptor level
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        sys.stderr.write(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: running with {world} ranks\n")
    host_wire = args.transport == "test"
    if host_wire:
        local_rank = 0          # every rank on GPU 0: the wire is host memory
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_shuffle
    ranks_seen = 1
    watchdog = Watchdog(args.peer_timeout, rank) if world > 1 else None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        import datetime
        if host_wire:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.peer_timeout))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=args.peer_timeout))

    if args.config != 3:
        import bench_configs
        bench_configs.run(args, world, rank, local_rank, json_fd, watchdog)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        if watchdog:
            watchdog.stop()
        os.close(json_fd)
        return

    from midoridb_amd.dev import DeviceCtx
    from midoridb_amd.dist import DistCtx, WIRE_32, WIRE_64

    t_ctx = time.perf_counter()
    dev = DeviceCtx(local_rank)
    dx = None
    if use_dist:
        # the exchange itself runs behind the C-ABI (include/mdb_dist.h: RCCL communicators created in C from an id that
        # travels through the launcher's process group); torch.distributed only provides the barrier and the reductions
        # of the timing contract
        dx = DistCtx.over_host_group(dev) if host_wire else DistCtx.from_torch(dev)
        ranks_seen = dx.allreduce_sum([1])[0]	# ranks that took part in a collective of the library's own communicator
        if ranks_seen != world:
            raise SystemExit(f"[bench] rank {rank}: the library's communicator saw {ranks_seen} ranks, the launcher {world}")

    def all_reduce_(t, op):
        """the timing contract's reductions: on the device over RCCL; through the host when the process group is gloo"""
        if host_wire:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    def beat(what):
        if watchdog:
            watchdog.beat(what)

    def make_tables(n_rank):
        """rank r holds rows [r*n, (r+1)*n) of the global tables (pre-sharded round-robin is equivalent for a permutation)"""
        total = n_rank * world
        mod = total // 16 if args.variant in ("D", "S") else 0
        b = dev.gen_keys(n_rank, rank * n_rank, total, 43, mod)
        if args.variant == "S":
            b.mul_(16)
        return (dev.gen_keys(n_rank, rank * n_rank, total, 42, 0), b, total, mod)

    def make_out(n_rank):
        cap = int(n_rank * 1.3) + 4096 if use_dist else n_rank
        return (torch.empty(cap, dtype=torch.int64, device=dev.device), torch.empty(cap, dtype=torch.int64, device=dev.device),
                torch.empty(cap, dtype=torch.int32, device=dev.device))

    def wire_format(a, b):
        """column statistics (computed once per table, outside the timed region, like a catalog would keep them): when every
        key of both columns fits 32 bits on every rank the exchange ships 4-byte keys"""
        if not use_dist or args.wire64:
            return False
        fits = 1.0
        for col in (a, b):
            lo, hi = dev.key_range(col)
            if lo > hi or lo < -(1 << 31) or hi >= (1 << 31):
                fits = 0.0
        t = torch.tensor([fits], dtype=torch.float64, device=dev.device)
        all_reduce_(t, dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def promise_ranges(a, b, w32):
        """the exchange handle is told the wire format and the two tables' GLOBAL key ranges (what a catalog keeps per column)"""
        if dx is None:
            return
        dx.set_wire(WIRE_32 if w32 else WIRE_64)
        # the same catalog statistics give every rank the two tables' GLOBAL key ranges: rows outside the other table's range
        # stay home (min-max pruning before the shuffle; a key outside its own promised range would be reported as an error)
        rng = []
        for col in (a, b):
            lo, hi = dev.key_range(col)
            t = torch.tensor([-lo, hi], dtype=torch.float64, device=dev.device)     # (keys < 2^53 here: exact in float64)
            all_reduce_(t, dist.ReduceOp.MAX)
            rng.append((-int(t[0].item()), int(t[1].item())))
        dx.set_key_ranges(rng[0], rng[1])

    def measure(n_rank, steps, warmup, cold=None):
        """warmup + timed loop over fresh tables of n_rank rows per rank -> dict (max over ranks, totals over ranks)"""
        a, b, total, mod = make_tables(n_rank)
        out = make_out(n_rank)
        w32 = wire_format(a, b)
        pipe = dx
        promise_ranges(a, b, w32)

        def step():
            if pipe is None and args.unordered:
                k, c, j = dev.join_group_count_unordered(a, None, b, None, out=out)
                return k.numel(), j
            if pipe is None:
                # (MDB_KEYS_MAY_ALIAS, as query_execute() calls the operator: when every left row is a group - variant U - the group keys ARE the
                # left key column and are not copied; the headline's variant D is not such a join: everything is written)
                k, c, f, j = dev.join_group_count(a, None, b, None, out=out, want_first=False, alias=True)
                return k.numel(), j
            k, c, j = pipe.join_group_count(a, None, b, None, out=out)
            return k.numel(), j

        if cold is not None:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            cold["cold_first_query_ms"] = (time.perf_counter() - t0) * 1e3
        g = j = 0
        for _ in range(max(warmup, 1) if world == 1 else warmup):
            beat("warm-up step")
            g, j = step()
        fault = os.environ.get("MDB_BENCH_FAULT", "")       # tests: `exit:<rank>` / `hang:<rank>` after the warm-up
        if fault and world > 1 and fault.split(":")[1] == str(rank):
            if fault.startswith("exit"):
                sys.stderr.write(f"[bench] rank {rank}: MDB_BENCH_FAULT=exit\n")
                os._exit(7)
            sys.stderr.write(f"[bench] rank {rank}: MDB_BENCH_FAULT=hang\n")
            while True:
                time.sleep(1)
        beat("barrier before the timed steps")
        barrier()
        c0 = dev.counters()
        t0 = time.perf_counter()
        for _ in range(steps):
            g, j = step()
        barrier()
        dt = time.perf_counter() - t0
        c1 = dev.counters()
        beat("timed steps done")
        # how the steps are distributed: the same number of steps again, each between two synchronisations of its own (`value` stays the mean
        # of the contract's loop above: one clock around all K steps)
        each = []
        for _ in range(steps):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            each.append((time.perf_counter() - t1) * 1e3)
        c2 = dev.counters()
        step_dist = dict(dist_of(each), retries_in_timed_steps=c1["retries"] - c0["retries"], samples_in_timed_steps=c1["samples"] - c0["samples"],
                         arena_grows_in_timed_steps=c1["arena_grows"] - c0["arena_grows"], alloc_misses_in_timed_steps=c1["alloc_misses"] - c0["alloc_misses"],
                         retries_in_distribution_steps=c2["retries"] - c1["retries"],
                         note="min / median / p90 / max over the same number of steps run right after the timed loop, each between its own "
                              "synchronisations (wall clock, this rank); *_in_timed_steps: mdb_dev_counters() read before and after the timed loop")
        red = torch.tensor([dt, float(j), float(g)], dtype=torch.float64, device=dev.device)
        if use_dist:
            tmax = red[:1].clone()
            all_reduce_(tmax, dist.ReduceOp.MAX)
            all_reduce_(red, dist.ReduceOp.SUM)
            red[0] = tmax[0]
        dt, joined, groups = float(red[0].item()), int(red[1].item()), int(red[2].item())
        return {"a": a, "b": b, "out": out, "pipe": pipe, "step": step, "wire32": w32, "mod": mod, "n": n_rank, "total_rows": total,
                "dt": dt, "ms_per_step": dt / steps * 1e3, "value": joined / (dt / steps), "joined": joined, "groups": groups, "step_ms": step_dist}

    cold = {}
    n_weak = args.rows
    n_strong = max(args.rows // world, 1)
    n = n_strong if (args.scaling == "strong" and world > 1) else n_weak
    m = measure(n, args.steps, args.warmup, cold)
    cold["context_and_first_query_ms"] = (time.perf_counter() - t_ctx) * 1e3
    a, b, out, step, pipeline = m["a"], m["b"], m["out"], m["step"], m["pipe"]
    dt, joined_total, groups_total, total_rows, mod_b, wire32 = m["dt"], m["joined"], m["groups"], m["total_rows"], m["mod"], m["wire32"]
    ms_per_step, value = m["ms_per_step"], m["value"]

    # ---- per-kernel profile (outside the timed region): live HIP events around every launch, on the launch stream
    dev.prof_enable(True)
    dev.prof_reset()
    prof_steps = 3
    for _ in range(prof_steps):
        step()
    prof = dev.prof_read()
    prof_syms = {k: dev.prof_symbols(k) for k in prof}     # rocprofv3's names of what ran under each profiler name
    dev.prof_enable(False)
    # the sharded operator, per rank: where a call's time goes (first level / the wire as far as it is not hidden / receiver) and
    # what the plan puts on ONE xGMI link per call, next to what that link can carry (MI355X: 7 links x ~153 GB/s per GPU, point to
    # point - an all-to-all's block for a peer crosses the one link to that peer)
    exchange = None
    beat("per-kernel profile done")
    if use_dist:
        dx.set_phase_timing(True)
        ph = []
        for _ in range(3):
            step()
            ph.append(dx.last_phases())
        dx.set_phase_timing(False)
        mine = {k: sorted(p[k] for p in ph)[1] for k in ph[0]}      # median of three
        plan = dx.last_plan() if dx.last_fused() else None
        per_rank = [None] * world
        if world > 1:
            dist.all_gather_object(per_rank, {"rank": rank, "phases_ms": mine})
        else:
            per_rank = [{"rank": 0, "phases_ms": mine}]
        XGMI_LINK_GBS = 153.0
        exchange = {"path": "first-level regions on the wire" if plan else "keys by destination",
                    "plan": plan, "per_rank": per_rank,
                    "phases_note": "HIP events on the operator's and the transfer stream: first_level_ms = this rank's partition passes; "
                                   "wire_wait_ms = how long after them the last block arrived (what of the transfer is NOT hidden behind the passes); "
                                   "receiver_ms = region descriptors, the receiver's own level if any, leaves; device_ms = all of it"}
        if plan:
            bpp = plan["bytes_per_peer"]
            exchange["per_link"] = {"bytes_per_peer_per_call": bpp, "peers": world - 1, "xgmi_link_GBs": XGMI_LINK_GBS,
                                    "predicted_link_ms": bpp / (XGMI_LINK_GBS * 1e9) * 1e3 if world > 1 else 0.0,
                                    "note": "bytes one rank sends to ONE peer per call (fixed-size blocks, slack included, + region counters) / one "
                                            "link's rate: the floor of wire time when every peer is one hop away and the links run in parallel; "
                                            "compare with first_level_ms (the transfer of table x overlaps the pass over table x + 1) and wire_wait_ms"}
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

    # the other scaling form, same process group, same kernels (N > 1 only)
    other = None
    if world > 1 and not args.no_secondary:
        n_other = n_weak if n == n_strong else n_strong
        if n_other != n:
            mo = measure(n_other, max(3, args.steps // 2), 1)
            other = {"scaling": "weak" if n_other == n_weak else "strong", "rows_per_table_per_gpu": n_other,
                     "rows_per_table_total": mo["total_rows"], "ms_per_step": mo["ms_per_step"], "value": mo["value"],
                     "joined_rows": mo["joined"], "groups": mo["groups"], "wire32": mo["wire32"]}
            for key in ("a", "b", "out", "pipe", "step"):
                mo.pop(key)

    # --verify at N > 1 (small tables): every rank's groups gathered on rank 0 and compared with the CPU oracle over the GLOBAL tables -
    # the groups are disjoint across ranks, their union in any order must be the oracle's
    verified_dist = None
    if args.verify and world > 1 and total_rows <= 20_000_000:
        beat("verify: one more step + gather of every rank's groups")
        promise_ranges(a, b, wire32)    # (the other scaling leg left ITS tables' ranges with the handle)
        k, c, jj = pipeline.join_group_count(a, None, b, None, out=out)
        parts = [None] * world
        dist.all_gather_object(parts, (k.cpu().numpy(), c.cpu().numpy(), int(jj)))
        if rank == 0:
            from oracle import cpu, np_oracle as orc
            eb = orc.gen_keys(total_rows, 0, total_rows, 43, mod_b) * (16 if args.variant == "S" else 1)
            ek, ec, ef, ej = cpu.hash_join_group_count(orc.gen_keys(total_rows, 0, total_rows, 42, 0), None, eb, None, os.cpu_count() or 1)
            gk, gc = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
            o1, o2 = np.argsort(gk, kind="stable"), np.argsort(ek, kind="stable")
            verified_dist = bool(sum(p[2] for p in parts) == ej and np.array_equal(gk[o1], ek[o2]) and np.array_equal(gc[o1], ec[o2]))

    if rank == 0:
        narrow = dev.last_join_narrow()
        pruned = dev.last_join_filter()[1]
        levels = 1 if (dev.last_join_levels() == 1 and os.environ.get("MDB_WORDS16", "1") != "0") else 2
        g_rank = groups_total / max(world, 1)
        kern = {k: {"launches_per_step": v[0] / prof_steps, "ms_per_step": v[1] / prof_steps, "rocprof_names": prof_syms.get(k, [])}
                for k, v in prof.items()}
        left_kept = g_rank if (pruned or (use_dist and args.variant == "D")) else n     # left rows that survive the range test (one per group in D)

        keys_alias = bool(dev.last_plan().get("keys_are_left_column")) if not use_dist else False

        def own_io(name):
            d = kern[name]
            return algorithmic_bytes(name, n, g_rank, narrow, pruned, levels, d["rocprof_names"], d["launches_per_step"], left_kept, keys_alias)
        for k, d in kern.items():   # per-kernel achieved rate on the bytes the kernel itself must move
            if d["ms_per_step"] > 0 and own_io(k) > 0:
                d["kernel_io_GBs"] = own_io(k) * d["launches_per_step"] / (d["ms_per_step"] * 1e-3) / 1e9
        pmc_variant = "shuffle" if use_dist else args.variant

        def roof_of(name):
            """`achieved` / `frac`: SURVEY 8(d)'s algorithmic bytes the launch consumes (8 B per row for a first-level pass, 16 B per
            group for the kernel that writes the result) / its average duration; kernels SURVEY 8(d) counts no bytes for are priced on
            their own I/O and say so (`basis`).  `frac_kernel_io`: the same launch on everything this design makes it read and write."""
            d = kern[name]
            launches = max(d["launches_per_step"], 1e-9)
            avg_ms = d["ms_per_step"] / launches
            io_bytes = own_io(name)
            sv = survey_bytes(name, n, g_rank)
            bytes_per_launch = sv if sv is not None else io_bytes
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            achieved_io = io_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            tr = pmc_traffic(d["rocprof_names"], pmc_variant) if (n == 100_000_000 and world == 1) else None
            return {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "basis": "SURVEY 8(d) bytes this launch consumes" if sv is not None else "kernel's own I/O (SURVEY 8(d) counts no bytes for this pass)",
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                    "kernel_io_bytes_per_launch": io_bytes, "achieved_kernel_io": achieved_io, "frac_kernel_io": achieved_io / HBM_PEAK_GBS,
                    "traffic": tr["bytes"] if tr else None, "traffic_source": tr["source"] if tr else None,
                    "traffic_over_kernel_io": (tr["bytes"] / io_bytes) if (tr and io_bytes) else None,
                    "d2d_copy_GBs": copy_gbs, "frac_of_d2d_copy": (achieved_io / copy_gbs) if copy_gbs else None,
                    "frac_of_d2d_copy_note": "kernel's own I/O over the box's copy rate (read + write): cannot exceed ~1",
                    "avg_launch_ms": avg_ms, "launches_per_step": d["launches_per_step"]}
        # dominant kernel = the one instance that takes the most time per step (one table per partition launch)
        dom_name = max(kern, key=lambda k: kern[k]["ms_per_step"]) if kern else None
        roof = roof_of(dom_name) if dom_name else None
        # whole-pipeline view: the bytes any correct algorithm must move once
        algo_bytes = 8 * 2 * total_rows + 16 * groups_total
        scaling_name = "strong" if (n == n_strong and world > 1) else "weak"
        line = {
            "metric": METRIC,
            "value": value, "unit": "joined rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling_name, "vs_baseline": None,
            "dtype": "int64", "data": "synthetic",
            "step_ms": m["step_ms"], "retries_in_timed_steps": m["step_ms"]["retries_in_timed_steps"], "last_plan": dev.last_plan(),
            "config": {"workload": f"A JOIN B ON id_a=id_b GROUP BY id_a COUNT(*), {n} rows/table/GPU x {world} GPU = {total_rows} rows/table "
                                   f"({scaling_name} scaling), variant {args.variant} "
                                   + {"D": "(B keys 16x duplicated, in the lowest sixteenth of A's key range)", "U": "(unique keys both sides)",
                                      "S": "(B keys 16x duplicated, spread over A's whole key range)"}[args.variant],
                       "key_form": ["wide (64-bit hashes)", "narrow (keys within one 2^32-wide window, verified on the device: 32-bit hashes)",
                                    "compact narrow (keys within the sampled 2^k-wide window, verified on the device: k-bit hashes, "
                                    "direct-address leaf tables)"][dev.last_join_form()],
                       "left_table_pruning": {"min_max": bool(dev.last_join_filter()[1]), "bitmap": int(dev.last_join_filter()[0])},
                       "partition_levels": dev.last_join_levels(),
                       "rows_per_table_per_gpu": n, "rows_per_table_total": total_rows, "joined_rows": joined_total, "groups": groups_total,
                       "result": "(group key, COUNT(*)) per group - the statement's two result columns, what query_execute() asks the operator for",
                       "operator_flags": "MDB_ORDER_FIRST | MDB_KEYS_MAY_ALIAS, as query_execute() calls the operator (group keys that would be an exact copy "
                                         "of the left key column - every left row a group: variant U - are not written; variant D and S: everything is written)",
                       "group_keys_are_left_column": bool(dev.last_plan().get("keys_are_left_column")) if not use_dist else False,
                       "order": "unspecified (--unordered: mdb_dev_join_group_count without MDB_ORDER_FIRST)" if (args.unordered and not use_dist) else
                                "reference first-occurrence order" if not use_dist else "per rank, unspecified (leaf order; first occurrence in the "
                                "received stream on the key-by-destination path)",
                       "parallelism": f"hash-partition x{world}, exchange behind the C-ABI (mdb_dist_join_group_count: RCCL all-to-all per table)"
                                      + (" (forced shuffle)" if args.force_shuffle and world == 1 else "")
                                      + ((", first-level partition regions on the wire (each table partitioned once; the receiver joins the "
                                          "regions of all ranks: mdb_dev_shard.hip)" if dx.last_fused() else
                                          (", 4-byte keys on the wire" if wire32 else ", 8-byte keys on the wire")) if use_dist else ""),
                       "rccl_ranks_seen": ("test transport" if host_wire else ranks_seen) if use_dist else None,
                       "transport": ("host memory through a gloo process group, all ranks on GPU 0 (--transport test: exercises this file's "
                                     "N > 1 code; the wire's timings mean nothing)" if host_wire else "RCCL") if use_dist else None,
                       "pruned_before_shuffle": bool(dx.last_pruned()) if use_dist else None},
            "roofline": roof,
            "pipeline": {"algorithmic_bytes": algo_bytes, "achieved_GBs": algo_bytes / (dt / args.steps) / 1e9,
                         "frac_of_peak": algo_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS / max(world, 1)},
            "kernels": kern,
        }
        if n == 100_000_000 and world == 1:
            line["pipeline"]["traffic"] = pmc_step_traffic(kern, pmc_variant, algo_bytes)
        if exchange is not None:
            line["exchange"] = exchange
        if world == 1:
            # what the sharded operator WOULD plan for this workload at 2, 4 and 8 GPUs (weak scaling: the same rows per GPU, the key
            # ranges of the N-fold tables) - mdb_dist_plan_preview, a host computation: which layout branch runs and what crosses one
            # xGMI link per call, so that the first multi-GPU run can be read against a prediction
            try:
                from midoridb_amd.dist import plan_preview
                prev = {}
                for w in (2, 4, 8):
                    tot = n * w
                    r_hi = {"D": tot // 16 - 1, "U": tot - 1, "S": 16 * (tot // 16 - 1)}[args.variant]
                    pp = plan_preview(w, [n, n], (0, tot - 1), (0, r_hi))
                    if pp is None:
                        prev[str(w)] = {"path": "keys by destination (the regions-on-the-wire plan does not serve this shape)"}
                        continue
                    prev[str(w)] = {"plan": {k: pp[k] for k in ("digit_bits", "digits_per_rank", "key_bits", "receiver_bits", "leaf_bits", "word_bytes")},
                                    "bytes_per_peer_per_call": pp["bytes_per_peer"],
                                    "predicted_link_ms": pp["bytes_per_peer"] / (153.0 * 1e9) * 1e3}
                line["exchange_preview"] = {"scaling": "weak", "xgmi_link_GBs": 153.0, "by_world": prev,
                                            "note": "predicted_link_ms = bytes one rank sends to ONE peer per call / one xGMI link's rate; the pass "
                                                    "over table x + 1 overlaps the transfer of table x, so the wire shows where it exceeds ~half the "
                                                    "first-level time (2 GPUs: every other row crosses the one link between the pair)"}
            except Exception as e:  # pragma: no cover
                line["exchange_preview"] = {"error": str(e)}
        # the first-level scatter of the left table beside it: the bandwidth-bound kernel of the pipeline
        for cand in ("part_scatter_l0", "part_scatter_l0_pruned", "part_scatter_l0_rid", "part_scatter_l0_w32"):
            if cand in kern and cand != dom_name:
                line["roofline_scatter"] = roof_of(cand)
                break
        if "roofline_scatter" not in line and use_dist and roof is not None:
            # the sharded operator's sender runs BOTH tables' first-level passes under one profiler name: that entry, averaged over the two
            first = [k for k in kern if k.startswith(("part_scatter_l0", "shard_scatter", "part_by_dest"))]
            if first:
                line["roofline_scatter"] = dict(roof_of(max(first, key=lambda k: kern[k]["ms_per_step"])),
                                                note="sharded operator: both tables' first-level passes run under this one profiler name")
        if other is not None:
            line[other["scaling"] + "_scaling"] = other
        elif world == 1:
            line["strong_scaling"] = {"same_as_primary": True, "note": "at N = 1 the weak and the strong workload are the same 2 x 10^8 rows"}
        line["cold_start"] = dict(cold, note="cold_first_query_ms: the first call in the process on fresh tables (scratch arena allocation, "
                                  "key sampling + its host sync, first-launch code load); context_and_first_query_ms adds the context creation; "
                                  "first_query_on_new_columns_ms: a first query over columns the context has never seen, with the catalog's "
                                  "statistics handed over as query_execute() does (mdb_dev_call_stats) - ..._sampled_ms: the same by a raw caller "
                                  "that passes none (key sample + host sync); `value` is the steady state of a repeated query")
        secondary = world == 1 and not use_dist and not args.no_secondary
        if secondary:
            reps = max(3, args.steps // 2)

            last_dist = {}

            def timed(fn):
                """mean seconds per call over `reps` calls, each between its own synchronisations (every operator call ends with a host read
                anyway); the distribution, the retries / arena growths / allocator misses the calls paid and the last call's plan are left
                in last_dist for the caller's line"""
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                c0 = dev.counters()
                each = []
                for _ in range(reps):
                    t1 = time.perf_counter()
                    r = fn()
                    torch.cuda.synchronize()
                    each.append((time.perf_counter() - t1) * 1e3)
                c1 = dev.counters()
                last_dist.clear()
                last_dist.update(dist_of(each), retries_in_timed_steps=c1["retries"] - c0["retries"], arena_grows_in_timed_steps=c1["arena_grows"] - c0["arena_grows"],
                                 alloc_misses_in_timed_steps=c1["alloc_misses"] - c0["alloc_misses"], plan=dev.last_plan())
                return sum(each) / reps * 1e-3, r
            try:
                # first query over columns the context has not seen (fresh pointers): pays the key sample and its sync, not the arena
                a2, b2 = a.clone(), b.clone()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                dev.join_group_count(a2, None, b2, None, out=out, want_first=False)
                torch.cuda.synchronize()
                line["cold_start"]["first_query_on_new_columns_sampled_ms"] = (time.perf_counter() - t1) * 1e3
                line["cold_start"]["first_query_on_new_columns_sampled_plan"] = {k: v for k, v in dev.last_plan().items() if k in ("samples", "retries", "from_stats")}
                del a2, b2
                # the same the way query_execute() runs it: the catalog's statistics of the key columns (kept by the store as rows are
                # ingested - here computed before the clock starts) handed to the operator (mdb_dev_call_stats): no key sample, no sync
                # for it, nothing remembered by address
                firsts = []
                for _ in range(5):      # (five pairs of fresh columns, each queried ONCE: the median - a single call's wall time is noisy)
                    a3, b3 = a.clone(), b.clone()
                    sa, sb = dev.key_range(a3), dev.key_range(b3)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    dev.call_stats(a3, sa, b3, sb)
                    dev.join_group_count(a3, None, b3, None, out=out, want_first=False)
                    dev.call_stats()
                    torch.cuda.synchronize()
                    firsts.append((time.perf_counter() - t1) * 1e3)
                    line["cold_start"]["first_query_on_new_columns_plan"] = {k: v for k, v in dev.last_plan().items() if k in ("samples", "retries", "from_stats")}
                    del a3, b3
                line["cold_start"]["first_query_on_new_columns_ms"] = sorted(firsts)[2]
                line["cold_start"]["first_query_on_new_columns_all_ms"] = firsts
            except Exception as e:  # pragma: no cover
                line["cold_start"]["error"] = str(e)
            try:
                # the wide form (64-bit hashes + row-id arrays), forced: what keys outside any 2^32 window run as
                dev.set_narrow_keys(0)
                dtw, rw = timed(lambda: dev.join_group_count(a, None, b, None, out=out, want_first=False))
                line["wide_form"] = {"ms_per_step": dtw * 1e3, "value": rw[3] / dtw, "narrow": dev.last_join_narrow(), "step_ms": dict(last_dist)}
            except Exception as e:  # pragma: no cover
                line["wide_form"] = {"error": str(e)}
            finally:
                dev.set_narrow_keys(1)
            def pipe_frac(groups, seconds, variant=None, fn=None):
                ab = 8 * 2 * n + 16 * groups
                d = {"algorithmic_bytes": ab, "achieved_GBs": ab / seconds / 1e9, "frac_of_peak": ab / seconds / 1e9 / HBM_PEAK_GBS}
                if variant is not None and fn is not None and n == 100_000_000:
                    # which kernels one step of THIS workload launches (live), priced with the committed PMC summary of the same workload
                    dev.prof_enable(True)
                    dev.prof_reset()
                    fn()
                    pr = dev.prof_read()
                    dev.prof_enable(False)
                    kk = {k: {"launches_per_step": float(v[0]), "rocprof_names": dev.prof_symbols(k)} for k, v in pr.items()}
                    d["traffic"] = pmc_step_traffic(kk, variant, ab)
                return d
            if "wide_form" in line and "ms_per_step" in line["wide_form"]:
                try:
                    dev.set_narrow_keys(0)
                    line["wide_form"]["pipeline"] = pipe_frac(groups_total, line["wide_form"]["ms_per_step"] * 1e-3, "wide",
                                                              lambda: dev.join_group_count(a, None, b, None, out=out, want_first=False))
                finally:
                    dev.set_narrow_keys(1)
            def unordered_of(b_tab, expect):
                # the same operator called without MDB_ORDER_FIRST and without first rows: the groups in unspecified order, no row ids
                # carried and no ordering sort (reported BESIDE the ordered figures, never as `value`)
                try:
                    dtx, rx = timed(lambda: dev.join_group_count_unordered(a, None, b_tab, None, out=out))
                    return {"ms_per_step": dtx * 1e3, "value": rx[2] / dtx, "step_ms": dict(last_dist), "served_by_unordered_form": bool(dev.last_join_unordered()),
                            "same_groups_and_joined_rows": bool(int(rx[0].numel()) == expect[0] and rx[2] == expect[1]),
                            "pipeline": pipe_frac(int(rx[0].numel()), dtx)}
                except Exception as e:  # pragma: no cover
                    return {"error": str(e)}
            line["unordered"] = unordered_of(b, (groups_total, joined_total))
            if args.variant == "D":
                # the unfavourable variants beside the headline, same pipeline: U (unique keys on both sides, G = n groups,
                # nothing to prune) and S (the headline's 16x duplication, but spread over A's whole key range: no range for min-max
                # pruning, no small key window - what a fact table whose keys are not bunched at the bottom of the dimension's looks like)
                for tag, workload, make_b in (
                        ("variant_U", f"unique keys both sides, {n} rows/table", lambda: dev.gen_keys(n, 0, n, 43, 0)),
                        ("variant_D_spread", f"B keys 16x duplicated and spread over A's whole range (id_b = 16 * (perm mod N/16)), {n} rows/table",
                         lambda: dev.gen_keys(n, 0, n, 43, n // 16).mul_(16))):
                    try:
                        b_x = make_b()
                        dtu, ru = timed(lambda: dev.join_group_count(a, None, b_x, None, out=out, want_first=False, alias=True))
                        plan_x = dev.last_plan()
                        line[tag] = {"workload": workload, "joined_rows": ru[3], "groups": int(ru[0].numel()), "ms_per_step": dtu * 1e3,
                                     "value": ru[3] / dtu, "plan": plan_x, "step_ms": dict(last_dist),
                                     "group_keys": ("the left key column itself (every left row is a group: MDB_KEYS_MAY_ALIAS, as query_execute() calls the "
                                                    "operator - not copied)") if plan_x.get("keys_are_left_column") else "written by the operator",
                                     "pipeline": pipe_frac(int(ru[0].numel()), dtu, "U" if tag == "variant_U" else "S",
                                                           lambda: dev.join_group_count(a, None, b_x, None, out=out, want_first=False)),
                                     "key_form": dev.last_join_form(), "partition_levels": dev.last_join_levels(),
                                     "min_max_pruning": bool(dev.last_join_filter()[1])}
                        if plan_x.get("keys_are_left_column"):
                            dtc, _ = timed(lambda: dev.join_group_count(a, None, b_x, None, out=out, want_first=False))
                            line[tag]["ms_per_step_keys_copied"] = dtc * 1e3      # the same call without the flag: out_key written (round 5's figure)
                        line[tag]["unordered"] = unordered_of(b_x, (int(ru[0].numel()), ru[3]))
                        del b_x
                    except Exception as e:  # pragma: no cover
                        line[tag] = {"error": str(e)}
            try:
                line["end_to_end"] = end_to_end(n, mod_b, a, b)
            except Exception as e:  # pragma: no cover
                line["end_to_end"] = {"error": str(e)}
        if args.no_cpu_baseline:
            line["cpu_baseline"] = None
            line["cpu_baseline_note"] = "--no-cpu-baseline"
        else:
            beat("CPU baseline (rank 0)")
            line["cpu_baseline"] = cpu_baseline()
            beat("CPU yardsticks (rank 0)")
            try:
                line["cpu_naive"] = cpu_naive_sizes()
            except Exception as e:  # pragma: no cover
                line["cpu_naive"] = {"error": str(e)}
            try:
                gpu_result = None
                if pipeline is None:
                    k, c, f, jj = dev.join_group_count(a, None, b, None, out=out, want_first=False)
                    gpu_result = (k, c, jj)
                line["cpu_hash"] = cpu_hash_yardstick(a, b, gpu_result)
            except Exception as e:  # pragma: no cover
                line["cpu_hash"] = {"error": str(e)}
        if args.verify and n <= 20_000_000 and world == 1:
            from oracle import cpu, np_oracle as orc
            eb = orc.gen_keys(n, 0, n, 43, mod_b) * (16 if args.variant == "S" else 1)
            ek, ec, ef, ej = cpu.hash_join_group_count(orc.gen_keys(n, 0, n, 42, 0), None, eb, None, os.cpu_count() or 1)
            if pipeline is None:
                k, c, f, jj = dev.join_group_count(a, None, b, None, out=out, want_first=False)
                ok = (jj == ej and np.array_equal(k.cpu().numpy(), ek) and np.array_equal(c.cpu().numpy(), ec))
            else:   # shuffled pipeline: same groups, order is not the reference's
                k, c, jj = pipeline.join_group_count(a, None, b, None, out=out)
                o1, o2 = np.argsort(k.cpu().numpy(), kind="stable"), np.argsort(ek, kind="stable")
                ok = (jj == ej and np.array_equal(k.cpu().numpy()[o1], ek[o2]) and np.array_equal(c.cpu().numpy()[o1], ec[o2]))
            line["verified_vs_oracle"] = bool(ok)
        if verified_dist is not None:
            line["verified_vs_oracle"] = verified_dist
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        beat("final barrier (rank 0 may be timing the CPU baseline)")
        dist.barrier()
        dx.close()
        dist.destroy_process_group()
    if watchdog:
        watchdog.stop()
    os.close(json_fd)


if __name__ == "__main__":
    main()
