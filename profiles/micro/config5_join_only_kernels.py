"""Which kernels the join-only form of BASELINE configs[4] runs (SELECT * over three 10^8-row tables on one key), and how long each takes:
    python profiles/micro/config5_join_only_kernels.py [rows]"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from midoridb_amd.query import DB
from midoridb_amd.dev import _bind, ProfEntry

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
SQL = sys.argv[2] if len(sys.argv) > 2 else "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c;"
with DB() as db:
    db.execute("CREATE TABLE A (id_a INT, x DOUBLE);")
    db.execute("CREATE TABLE B (id_b INT, y DOUBLE);")
    db.execute("CREATE TABLE C (id_c INT, z INT);")
    for t, seed in (("A", 42), ("B", 43), ("C", 44)):
        db.generate_shard(t, n, 0, n, seed, [0, 0])
    db.results_on_device(True)
    lib, h = db.lib, db.device_handle()
    _bind(lib)
    for _ in range(2):
        db.query_device(SQL, copy=False)
    lib.mdb_dev_prof_enable(h, 1)
    lib.mdb_dev_prof_reset(h)
    t0 = time.perf_counter()
    for _ in range(3):
        r = db.query_device(SQL, copy=False)
    lib.mdb_dev_sync(h)
    wall = (time.perf_counter() - t0) / 3 * 1e3
    buf = (ProfEntry * 64)()
    cnt = ctypes.c_int()
    lib.mdb_dev_prof_read(h, buf, 64, ctypes.byref(cnt))
    kern = {buf[i].name.decode(): [buf[i].launches / 3, round(buf[i].total_ms / 3, 4)] for i in range(cnt.value)}
    lib.mdb_dev_prof_enable(h, 0)
    t0 = time.perf_counter()
    for _ in range(3):
        r = db.query_device(SQL, copy=False)
    lib.mdb_dev_sync(h)
    wall2 = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps({"rows": n, "query": SQL, "result_rows": r[3], "ms_per_statement_profiled": round(wall, 3), "ms_per_statement": round(wall2, 3),
                      "kernel_ms_sum": round(sum(v[1] for v in kern.values()), 3),
                      "kernels": dict(sorted(kern.items(), key=lambda kv: -kv[1][1]))}))
