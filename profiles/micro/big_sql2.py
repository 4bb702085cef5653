"""BASELINE-config shaped statements through query_execute() on device-generated tables (unique keys, INT payload (generated tables hold INT columns))."""
import sys, time
sys.path.insert(0, '.')
from midoridb_amd.query import DB
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
db = DB()
db.execute("CREATE TABLE A (id_a INT, x INT);")
db.execute("CREATE TABLE B (id_b INT, y INT);")
db.execute("CREATE TABLE C (id_c INT, z INT);")
db.generate("A", n, 42, [0, 0])
db.generate("B", n, 43, [0, 0])
db.generate("C", n, 44, [0, 1000])
stmts = [
    "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b LIMIT 10;",
    "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c LIMIT 10;",
    "SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c GROUP BY id_a LIMIT 10;",
    "SELECT z, COUNT(*) FROM A INNER JOIN C ON A.id_a = C.id_c GROUP BY z;",
    "SELECT COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c WHERE z < 10;",
    "SELECT id_a, x, y FROM A INNER JOIN B ON A.id_a = B.id_b WHERE x < 1000 LIMIT 10;",
    "SELECT id_a FROM A ORDER BY x LIMIT 10;",
    "SELECT DISTINCT z FROM C;",
]
for q in stmts:
    try:
        db.query(q)
        t0 = time.perf_counter()
        r = db.query(q)
        wall = (time.perf_counter() - t0) * 1e3
        print(f"{r.exec_ms:9.3f} ms exec {wall:9.3f} ms wall  rows {r.nrows:>10}  {q}")
    except Exception as e:
        print("ERROR", q, str(e)[:300])
db.close()
