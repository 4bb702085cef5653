"""BASELINE configs[4]'s join-only form on one GPU - SELECT * FROM A JOIN B ON id_a = id_b JOIN C ON id_a = id_c with x, y DOUBLE and z INT
carried, 10^8 unique keys per table, results kept on the device - with the per-kernel time of every statement (MDB_PROF_DUMP=1).
Round 4: the payload of B and C travels through their partition levels (mdb_dev_join_payload, two levels for the 2^27-value window)
instead of pairs + ordering sort + random gathers."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MDB_PROF_DUMP", "1")
from midoridb_amd.query import DB  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
with DB() as db:
    db.execute("CREATE TABLE A (id_a INT, x DOUBLE);")
    db.execute("CREATE TABLE B (id_b INT, y DOUBLE);")
    db.execute("CREATE TABLE C (id_c INT, z INT);")
    for t, seed in (("A", 42), ("B", 43), ("C", 44)):
        db.generate_shard(t, n, 0, n, seed, [0, 0])
    db.results_on_device(True)
    q = "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c;"
    for i in range(3):
        r = db.query_device(q, copy=False)
        print("rows", r[3], "executor ms", round(r[5], 3), flush=True)
