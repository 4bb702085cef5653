"""Bulk ingest of two 10^8-row INT columns from host arrays through mdb_table_append_columns (the drop-in API), device context and arena in
place before the clock starts; MDB_INGEST_THP=0|1 (huge pages for the host store), MDB_INGEST_THREADS:
    python profiles/micro/ingest.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from midoridb_amd.query import DB
from midoridb_amd.dev import _bind as _bind_dev
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ha = np.random.default_rng(1).permutation(n).astype(np.int64)
hb = (ha[::-1] // 16).copy()
with DB() as db:
    db.execute("CREATE TABLE A (id_a INT);")
    db.execute("CREATE TABLE B (id_b INT);")
    _bind_dev(db.lib)
    db.lib.mdb_dev_reserve(db.device_handle(), 10 << 30)
    t0 = time.perf_counter()
    db.append_columns("A", [ha])
    db.append_columns("B", [hb])
    ms = (time.perf_counter() - t0) * 1e3
    r = db.query("SELECT id_a, COUNT(*) FROM A INNER JOIN B ON id_a = id_b GROUP BY id_a;")
    print("THP %s threads %s: ingest %.1f ms, first SELECT %.2f ms, %d groups" % (os.environ.get("MDB_INGEST_THP", "default"), os.environ.get("MDB_INGEST_THREADS", "default"),
                                                                         ms, db.last_call_ms, len(r.columns[0])))
