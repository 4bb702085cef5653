#!/usr/bin/env python3
"""bench_operators.py - per-operator measurements of the OTHER rows of the hot path, beside the
headline pipeline that bench.py times: scan+filter (config 1 scaled up), materialising join with payload
(config 2), single-table GROUP BY, three-way join.  Not part of the driver's bench contract; writes one
JSON document (default profiles/r02/operators.json) with milliseconds and the rate on each operator's
algorithmic bytes, all measured with HIP events through the library's per-kernel profiler.

    python bench_operators.py [--out profiles/r02/operators.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from midoridb_amd import dev as D  # noqa: E402
from midoridb_amd.dev import DeviceCtx  # noqa: E402


def timed(dev, fn, reps=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    dev.prof_enable(True)
    dev.prof_reset()
    fn()
    prof = dev.prof_read()
    dev.prof_enable(False)
    return ms, {k: round(v[1], 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:8]}, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04", "operators.json"))
    ap.add_argument("--configs1", action="store_true", help="only the two BASELINE configs[0..1] shapes (scan_filter_1e8, join_payload_1e7): "
                                                             "what profiles/collect.sh runs under rocprofv3")
    args = ap.parse_args()
    dev = DeviceCtx(0)
    res = {}

    # ---- scan + WHERE (config 1 shape at 10^8 rows): SELECT v FROM T WHERE v > N/2 ; v = permutation
    n = 100_000_000
    v = dev.gen_keys(n, 0, n, 7, 0)
    prog = [(D.P_CMP_COL_CONST, D.CMP_GT, D.T_INT64, 0, 0, n // 2)]

    def scan_filter():
        # scan + WHERE + projection in one operator and ONE pass (mdb_dev_filter_project): the kernel that evaluates the
        # predicate writes the survivors at their final positions (decoupled look-back); the output tensor is the
        # library's buffer itself (no copy)
        m, _ = dev.filter_project(prog, [(v, None, None)], n, [(v, None)])
        return m
    ms, kern, m = timed(dev, scan_filter)

    def scan_filter_unfused():
        sel = dev.filter(prog, [(v, None, None)], n)
        out, _ = dev.gather64(v, None, sel, sel.numel())
        return sel.numel()
    ms_unfused, kern_unfused, _ = timed(dev, scan_filter_unfused)
    algo = 8 * n + 8 * m            # read every value once, write the survivors once
    res["scan_filter_1e8"] = {"rows_in": n, "rows_out": m, "ms": ms, "algorithmic_bytes": algo,
                              "algorithmic_GBs": algo / (ms * 1e-3) / 1e9, "kernels_ms": kern,
                              "device_ms": sum(kern.values()), "device_algorithmic_GBs": algo / (sum(kern.values()) * 1e-3) / 1e9,
                              "unfused_ms": ms_unfused, "unfused_kernels_ms": kern_unfused,
                              "note": "mdb_dev_filter_project: one pass, survivors written at their final positions (decoupled look-back over "
                                      "16384-row blocks); 50% selectivity; ms = wall time per call incl. the host synchronisation that returns the "
                                      "row count, device_ms = the operator's kernels; unfused = filter (bitmap, scan, positions) + projection "
                                      "gather, the round-1 path"}

    # ---- materialising join with payload (config 2): 10^7 x 10^7, 1:1 keys, 4 output columns
    n2 = 10_000_000
    a_id, a_f = dev.gen_keys(n2, 0, n2, 42, 0), dev.gen_keys(n2, 0, n2, 43, 0)
    b_id, b_f = dev.gen_keys(n2, 0, n2, 50, 0), dev.gen_keys(n2, 0, n2, 51, 0)

    def join_payload():
        l, r = dev.join_pairs(a_id, None, b_id, None)
        j = l.numel()
        # the projection as the executor plans it (mdb_exec.c, projection pass 1): B's join key holds A's value in every joined
        # tuple, so both key columns of SELECT * are ONE gather of A's column through the left row ids (ascending: near-sequential
        # reads); the payload columns are gathered through their own row ids; one launch
        # ... and when every left row found exactly one partner (mdb_dev_last_pairs_identity: the primary-key join) the left row ids
        # are 0, 1, 2 ...: A's columns are read as they stand
        lid = None if dev.last_pairs_identity() else l
        dev.gather_cols([(a_id, None, lid), (a_f, None, lid), (b_f, None, r)], j)
        return j
    ms_pairs, kern_pairs, j = timed(dev, join_payload, reps=3, warmup=1)
    algo = 8 * 2 * n2 + 8 * 2 * n2 + 8 * 4 * j      # keys + payload columns read once, 4 result columns written

    def join_payload_carried():
        # the executor's plan since round 4 (mdb_exec.c: join_with_payload): every left row has its one partner, so B's payload cell
        # travels through B's one partition level and lands at the left row's place - no partner row ids, no compaction, no random
        # gather; A's columns are read as they stand (copied: the result owns its columns), B's key column is A's
        out = dev.join_payload(a_id, None, b_id, None, [b_f])
        if out is None:
            return join_payload()
        dev.gather_cols([(a_id, None, None), (a_f, None, None)], n2)
        return n2
    ms, kern, j = timed(dev, join_payload_carried, reps=3, warmup=1)
    res["join_payload_1e7"] = {"rows_per_table": n2, "joined_rows": j, "ms": ms, "joined_rows_per_s": j / (ms * 1e-3),
                               "algorithmic_bytes": algo, "algorithmic_GBs": algo / (ms * 1e-3) / 1e9, "frac_of_peak": algo / (ms * 1e-3) / 1e9 / 8000.0,
                               "kernels_ms": kern,
                               "note": "mdb_dev_join_payload: the right table's payload cell carried through its one partition level and written at "
                                       "the left row's place (every left row has its one partner: verified on the device) + the copies of A's "
                                       "two columns into the result; B's key column is A's",
                               "pairs_path": {"ms": ms_pairs, "kernels_ms": kern_pairs,
                                              "note": "round 3's plan: mdb_dev_join_pairs (pairs in the reference's (l, r) order) + the projection in "
                                                      "one launch (mdb_dev_gather_cols): 3 gathers for the 4 result columns"}}

    def join_payload4():
        l, r = dev.join_pairs(a_id, None, b_id, None)
        j = l.numel()
        dev.gather_cols([(a_id, None, l), (b_id, None, r), (a_f, None, l), (b_f, None, r)], j)
        return j
    ms, kern, j = timed(dev, join_payload4, reps=3, warmup=1)
    res["join_payload_1e7_four_gathers"] = {"rows_per_table": n2, "joined_rows": j, "ms": ms, "kernels_ms": kern,
                                            "note": "the same with every result column gathered on its own (the round-1 plan)"}
    if args.configs1:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)
        print(json.dumps({k: {kk: vv for kk, vv in d.items() if kk != "kernels_ms"} for k, d in res.items()}, indent=1))
        return

    # ---- the row-order payload join at 10^8 x 10^8 rows (mdb_dev_rowjoin.hip; BASELINE configs[4]'s join-only shape at operator level): one right
    #      table with one and two cells, and B and C in ONE call (mdb_dev_join_payload_multi: the left table's tiles sorted once)
    ka, kb, kc = dev.gen_keys(n, 0, n, 42, 0), dev.gen_keys(n, 0, n, 43, 0), dev.gen_keys(n, 0, n, 44, 0)
    pb, pc = kb * 3 + 1, kc * 5 - 2
    for name, fn, cols in (("join_payload_1e8_one_cell", lambda: dev.join_payload(ka, None, kb, None, [pb])[0].numel(), 1),
                           ("join_payload_1e8_two_cells", lambda: dev.join_payload(ka, None, kb, None, [pb, kb])[0].numel(), 2),
                           ("join_payload_multi_1e8_two_tables", lambda: dev.join_payload_multi(ka, [(kb, [pb]), (kc, [pc])], 0, n - 1)[0][0].numel(), 2)):
        ms, kern, j = timed(dev, fn, reps=3, warmup=1)
        tables = 2 if "multi" in name else 1
        algo = 8 * n + tables * 8 * n + cols * 8 * n + cols * 8 * j         # key columns and payload columns read once, the carried columns written
        res[name] = {"rows_per_table": n, "joined_rows": j, "carried_columns": cols, "ms": ms, "joined_rows_per_s": j / (ms * 1e-3), "algorithmic_bytes": algo,
                     "frac_of_peak": algo / (ms * 1e-3) / 1e9 / 8000.0, "kernels_ms": kern, "payload_tables": dev.last_plan()["payload_tables"],
                     "note": "every left row has its one partner in every right table (verified on the device); the carried columns in the left table's row order"}
    got = dev.join_payload_multi(ka, [(kb, [pb]), (kc, [pc])], 0, n - 1)
    res["join_payload_multi_1e8_two_tables"]["identical_to_f_of_key"] = bool(torch.equal(got[0][0], ka * 3 + 1) and torch.equal(got[1][0], ka * 5 - 2))
    del ka, kb, kc, pb, pc, got
    torch.cuda.empty_cache()

    # ---- single-table GROUP BY key COUNT(*) at 10^8 rows, 6.25M groups of 16
    keys = dev.gen_keys(n, 0, n, 43, n // 16)

    def group_by():
        f, c = dev.group_count(keys, None)
        return f.numel()
    ms, kern, g = timed(dev, group_by)
    algo = 8 * n + 12 * g
    res["group_count_1e8"] = {"rows": n, "groups": g, "ms": ms, "algorithmic_bytes": algo,
                              "algorithmic_GBs": algo / (ms * 1e-3) / 1e9, "kernels_ms": kern}
    gk_out = (torch.empty(n, dtype=torch.int64, device=dev.device), torch.empty(n, dtype=torch.int64, device=dev.device))

    def group_by_keys():
        r = dev.group_count_keys(keys, None, out=gk_out)
        return -1 if r is None else r[0].numel()
    ms, kern, gk = timed(dev, group_by_keys)
    algo = 8 * n + 16 * max(gk, 0)
    res["group_count_keys_any_order_1e8"] = {"rows": n, "groups": gk, "ms": ms, "algorithmic_bytes": algo, "algorithmic_GBs": algo / (ms * 1e-3) / 1e9,
                                            "kernels_ms": kern, "note": "mdb_dev_group_count_keys: (key, COUNT) pairs in unspecified order - no row "
                                                                          "ids, no ordering sort (mdb_database_groups_any_order)"}
    del gk_out

    # ---- three-way join on one key (config 5 shape), 10^7 rows per table, 1:1:1
    c_id = dev.gen_keys(n2, 0, n2, 60, 0)

    def three_way():
        l, r = dev.join_pairs(a_id, None, b_id, None)
        k_ab, _ = dev.gather64(a_id, None, l, l.numel())     # key column of the joined stream
        p, q = dev.join_pairs(k_ab, None, c_id, None)
        ra = dev.gather32(l, p)                               # compose row ids: A, B through the second join
        rb = dev.gather32(r, p)
        return p.numel()
    ms, kern, j3 = timed(dev, three_way, reps=3, warmup=1)
    res["three_way_join_1e7"] = {"rows_per_table": n2, "joined_rows": j3, "ms": ms, "joined_rows_per_s": j3 / (ms * 1e-3),
                                 "kernels_ms": kern}

    # ---- BASELINE config 5 shape on one GPU at full size: (A join B) join C on one key, 10^8 rows per table, then
    #      GROUP BY the key + COUNT(*); row ids composed through both joins (what a payload projection would gather with)
    del a_id, a_f, b_id, b_f, c_id
    torch.cuda.empty_cache()
    big = [dev.gen_keys(n, 0, n, s, 0) for s in (42, 43, 44)]

    def three_way_group():
        l, r = dev.join_pairs(big[0], None, big[1], None)
        k_ab, _ = dev.gather64(big[0], None, l, l.numel())
        p, q = dev.join_pairs(k_ab, None, big[2], None)
        ra, rb = dev.gather32(l, p), dev.gather32(r, p)
        key, _ = dev.gather64(k_ab, None, p, p.numel())
        first, cnt = dev.group_count(key, None)
        return first.numel()
    ms, kern, g3 = timed(dev, three_way_group, reps=2, warmup=1)
    res["three_way_join_group_1e8"] = {"rows_per_table": n, "groups": g3, "ms": ms, "joined_rows_per_s": n / (ms * 1e-3), "kernels_ms": kern,
                                       "note": "BASELINE configs[4] shape on ONE GPU: two materialising joins (unique keys) + GROUP BY + COUNT(*)"}

    def three_way_fused():
        k, c, f, j = dev.join_group_count_multi(big[0], None, [(big[1], None), (big[2], None)])
        return k.numel()
    ms, kern, g3f = timed(dev, three_way_fused, reps=3, warmup=1)
    res["three_way_join_group_fused_1e8"] = {"rows_per_table": n, "groups": g3f, "ms": ms, "joined_rows_per_s": n / (ms * 1e-3), "kernels_ms": kern,
                                             "one_pass": dev.last_join_multi(),
                                             "note": "the same query through mdb_dev_join_group_count_multi: every table partitioned once, the right "
                                                     "tables' counts multiplied in the leaf kernel, the groups ordered once; no joined row exists"}

    def three_way_unordered():
        k, c, j = dev.join_group_count_multi_unordered(big[0], None, [(big[1], None), (big[2], None)])
        return k.numel()
    ms, kern, g3u = timed(dev, three_way_unordered, reps=3, warmup=1)
    res["three_way_join_group_unordered_1e8"] = {"rows_per_table": n, "groups": g3u, "ms": ms, "joined_rows_per_s": n / (ms * 1e-3), "kernels_ms": kern,
                                                 "unordered_form": dev.last_join_unordered(),
                                                 "note": "the same call without MDB_ORDER_FIRST (mdb_database_groups_any_order): no row ids, no ordering sort"}
    del big
    torch.cuda.empty_cache()

    # ---- ORDER BY (extension): stable sort permutation of 10^8 rows on one INT64 key with 27
    #      significant bits (value range + stream position fit one word: range pass, pack pass, two scatter levels, a
    #      per-leaf LDS sort), the same with two key columns, and a top-k over the north-star groups through query_execute()
    k1 = dev.gen_keys(n, 0, n, 77, 0)
    k2 = dev.gen_keys(n, 0, n, 78, 512)

    def order_by():
        return dev.sort_perm([(k1, None, None, D.T_INT64, False)], n).numel()
    ms, kern, _ = timed(dev, order_by, reps=3, warmup=1)
    algo = n * 8 + n * (8 + 8) + 2 * n * (8 + 8) + n * (8 + 4)  # range read; pack read + write; two scatter levels; leaf read + perm out
    res["order_by_1e8"] = {"rows": n, "ms": ms, "rows_per_s": n / (ms * 1e-3), "moved_bytes": algo,
                           "moved_GBs": algo / (ms * 1e-3) / 1e9, "kernels_ms": kern,
                           "note": "mdb_dev_sort_perm, permutation keys < 2^27: packed-word path (the stable 8-bit LSD passes it "
                                   "replaces took 4.6 ms); moved_bytes is what this method moves, not a lower bound"}

    def order_by_limit():
        return dev.topk_perm([(k1, None, None, D.T_INT64, False)], n, 10)[1]
    ms, kern, cand = timed(dev, order_by_limit, reps=3, warmup=1)
    res["order_by_limit10_1e8"] = {"rows": n, "k": 10, "rows_sorted": cand, "ms": ms, "rows_per_s": n / (ms * 1e-3),
                                   "algorithmic_GBs": n * 8 / (ms * 1e-3) / 1e9, "kernels_ms": kern,
                                   "note": "mdb_dev_topk_perm: sample threshold + one filter pass + sort of the candidates, against order_by_1e8's "
                                           "full sort; algorithmic bytes = one read of the key column"}

    def order_by2():
        return dev.sort_perm([(k2, None, None, D.T_INT64, True), (k1, None, None, D.T_INT64, False)], n).numel()
    ms, kern, _ = timed(dev, order_by2, reps=3, warmup=1)
    res["order_by_two_keys_1e8"] = {"rows": n, "ms": ms, "rows_per_s": n / (ms * 1e-3), "kernels_ms": kern,
                                    "note": "ORDER BY k2 DESC, k1 ASC (9 + 27 value bits + 27 position bits = 63 bits: one word)"}
    # GROUP BY a column of 1000 distinct values (the direct LDS-table path) - the other common shape of the operator
    def group_by_small():
        return dev.group_count(k2, None)[0].numel()
    ms, kern, gsmall = timed(dev, group_by_small, reps=5, warmup=2)
    res["group_count_small_range_1e8"] = {"rows": n, "groups": gsmall, "ms": ms, "rows_per_s": n / (ms * 1e-3),
                                          "algorithmic_GBs": n * 8 / (ms * 1e-3) / 1e9, "kernels_ms": kern,
                                          "note": "GROUP BY over 512 distinct values + COUNT(*): one streaming pass, per-workgroup LDS tables "
                                                  "(3.8 ms through the partitioned path's hot-key kernels)"}

    # GROUP BY two columns + COUNT(*) (composite-key semantics; round 6: the columns' composite value as one key column through the
    # single-column operator where the ranges fit 63 bits together - sort, run heads, run lengths before and otherwise)
    k3 = dev.gen_keys(n, 0, n, 79, 300)

    def group_by2():
        return dev.group_count_multi([(k2, None, None, D.T_INT64, False), (k3, None, None, D.T_INT64, False)], n)[0].numel()
    ms, kern, groups2 = timed(dev, group_by2, reps=3, warmup=1)
    res["group_by_two_columns_1e8"] = {"rows": n, "groups": groups2, "ms": ms, "rows_per_s": n / (ms * 1e-3), "kernels_ms": kern,
                                       "note": "GROUP BY k2, k3 (512 x 300 value combinations) + COUNT(*), first-occurrence order"}

    def distinct2():
        return dev.distinct_sel([(k2, None, None, D.T_INT64, False), (k3, None, None, D.T_INT64, False)], n).numel()
    ms, kern, groups2 = timed(dev, distinct2, reps=3, warmup=1)
    res["distinct_two_columns_1e8"] = {"rows": n, "distinct": groups2, "ms": ms, "rows_per_s": n / (ms * 1e-3), "kernels_ms": kern,
                                       "note": "SELECT DISTINCT k2, k3: first rows of the combinations, ascending"}
    del k3
    del k2
    del k1

    # ---- the north-star query end to end through the drop-in C API (query_execute), tables generated on the
    #      device: includes planning, the device pipeline and the D2H copy of the result columns
    from midoridb_amd.query import DB
    del v, keys
    torch.cuda.empty_cache()
    with DB() as db:
        # BASELINE configs[1] through the drop-in API: 10^7 x 10^7 rows, 4 result columns (320 MB of result over PCIe)
        db.execute("CREATE TABLE A2 (id_a2 INT, f1 INT);")
        db.execute("CREATE TABLE B2 (id_b2 INT, f2 INT);")
        db.generate("A2", n2, 42)
        db.generate("B2", n2, 50)
        q2 = "SELECT * FROM A2 INNER JOIN B2 ON A2.id_a2 = B2.id_b2;"
        db.query(q2)
        r2 = db.query(q2)
        res["join_payload_via_query_execute_1e7"] = {"rows_per_table": n2, "joined_rows": r2.nrows, "executor_ms": r2.exec_ms, "call_ms": db.last_call_ms,
                                                     "note": "query_execute(): plan + join + projection + D2H of 4 x 10^7 x 8 B into pinned host columns"}
    with DB() as db:
        db.execute("CREATE TABLE A (id_a INT);")
        db.execute("CREATE TABLE B (id_b INT);")
        db.generate("A", n, 42)
        db.generate("B", n, 43, [n // 16])
        q = "SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a;"
        db.query(q)
        t0 = time.perf_counter()
        r = db.query(q)
        wall = (time.perf_counter() - t0) * 1e3
        call = db.last_call_ms
        qk = ("SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a HAVING COUNT(*) > 1 "
              "ORDER BY id_a DESC LIMIT 10;")
        db.query(qk)
        t1 = time.perf_counter()
        rk = db.query(qk)
        res["north_star_top10_via_query_execute_1e8"] = {"rows_per_table": n, "result_rows": rk.nrows, "executor_ms": rk.exec_ms,
                                                         "call_ms": db.last_call_ms, "python_wall_ms": (time.perf_counter() - t1) * 1e3,
                                                         "note": "fused join+group count, HAVING filter, ORDER BY (radix sort of 6.25M groups), LIMIT"}
        # BASELINE configs[4] shape with aggregation: A JOIN B JOIN C on one key + GROUP BY + COUNT(*): the fused operator is
        # chained, none of the 1.6*10^9 joined rows is materialised
        db.execute("CREATE TABLE C (id_c INT);")
        db.generate("C", n, 44, [n // 16])
        q3 = "SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c GROUP BY id_a;"
        db.query(q3)
        t2 = time.perf_counter()
        r3 = db.query(q3)
        res["three_way_fused_via_query_execute_1e8"] = {
            "rows_per_table": n, "groups": r3.nrows, "joined_rows": r3.joined_rows, "executor_ms": r3.exec_ms,
            "call_ms": db.last_call_ms, "python_wall_ms": (time.perf_counter() - t2) * 1e3,
            "joined_rows_per_s": r3.joined_rows / (r3.exec_ms * 1e-3),
            "note": "chained fused join + group count (B and C hold every key < n/16 sixteen times: COUNT(*) = 256 per group); "
                    "executor_ms includes the D2H of the 6.25 M result rows"}
        res["north_star_via_query_execute_1e8"] = {
            "rows_per_table": n, "groups": r.nrows, "joined_rows": r.joined_rows, "executor_ms": r.exec_ms, "call_ms": call,
            "python_wall_ms": wall, "joined_rows_per_s_call": r.joined_rows / (call * 1e-3),
            "note": "query_execute() on device-resident tables; executor_ms = plan + device pipeline + D2H of the result "
                    "(2 columns x G x 8 B over PCIe); call_ms = wall time of the C call (query_execute) alone; python_wall_ms adds "
                    "the binding's copy of the result columns into numpy arrays (not part of the C API)"}

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in d.items() if kk != "kernels_ms"} for k, d in res.items()}, indent=1))


if __name__ == "__main__":
    main()
