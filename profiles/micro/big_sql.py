"""Executor time of a set of large statements through query_execute() on device-generated tables (10^8 rows unless
told otherwise): looking for operators that are out of line with the bytes they have to move."""
import sys, time
sys.path.insert(0, '.')
from midoridb_amd.query import DB
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
db = DB()
db.execute("CREATE TABLE A (id_a INT, fa INT);")
db.execute("CREATE TABLE B (id_b INT, fb INT);")
db.generate("A", n, 42, [0, 1000])          # id_a permutation, fa = permutation mod 1000
db.generate("B", n, 43, [n // 16, 300])
stmts = [
    "SELECT COUNT(*) FROM A WHERE fa > 500;",
    "SELECT id_a FROM A WHERE fa = 7;",
    "SELECT fa, COUNT(*) FROM A GROUP BY fa;",
    "SELECT DISTINCT fa FROM A;",
    "SELECT fa, COUNT(*) FROM A GROUP BY fa HAVING COUNT(*) > 100000 ORDER BY fa DESC LIMIT 5;",
    "SELECT fb, id_b, COUNT(*) FROM B GROUP BY fb, id_b LIMIT 10;",
    "SELECT DISTINCT fb, id_b FROM B LIMIT 10;",
    "SELECT id_a, fa FROM A ORDER BY fa, id_a LIMIT 10;",
    "SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a ORDER BY id_a LIMIT 10;",
    "SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b WHERE fa < 100 GROUP BY id_a LIMIT 10;",
    "SELECT COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b;",
    "SELECT fa, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY fa;",
    "SELECT fb, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b WHERE fa < 500 AND fb > 100 GROUP BY fb;",
    "SELECT id_a, fa, fb FROM A INNER JOIN B ON A.id_a = B.id_b WHERE fa = 3 AND fb = 5;",
    "SELECT id_a, fa FROM A WHERE fa IN (1, 2, 3) LIMIT 10;",
    "SELECT fa FROM A ORDER BY fa DESC LIMIT 10;",
    "SELECT fb, id_b FROM B ORDER BY id_b, fb LIMIT 10;",
    "SELECT COUNT(*) FROM A;",
    "UPDATE A SET fa = 5 WHERE fa = 6;",
    "DELETE FROM B WHERE fb = 299;",
]
for q in stmts:
    try:
        if q.startswith("SELECT"):
            db.query(q)
            t0 = time.perf_counter()
            r = db.query(q)
            wall = (time.perf_counter() - t0) * 1e3
            print(f"{r.exec_ms:9.3f} ms exec {wall:9.3f} ms wall  rows {r.nrows:>10}  {q}")
        else:
            t0 = time.perf_counter()
            aff = db.execute(q)
            print(f"{(time.perf_counter() - t0) * 1e3:9.3f} ms wall                 affected {aff}  {q}")
    except Exception as e:
        print("ERROR", q, str(e)[:200])
db.close()
