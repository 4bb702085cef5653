"""More statement shapes of the reference's path: cross joins, non-equi ON, NULL predicates, IN lists, XOR."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from midoridb_amd.query import DB
db = DB()
db.execute("CREATE TABLE S (ks INT, vs INT);")
db.execute("CREATE TABLE T (kt INT, vt INT);")
db.execute("CREATE TABLE A (id_a INT, fa INT);")
db.execute("CREATE TABLE B (id_b INT, fb INT);")
rng = np.random.default_rng(1)
m = 10_000
db.append_columns("S", [rng.integers(0, 1000, m), rng.integers(0, 100, m)], [None, rng.random(m) < 0.1])
db.append_columns("T", [rng.integers(0, 1000, m), rng.integers(0, 100, m)], [None, None])
n = 30_000_000
db.generate("A", n, 42, [0, 1000])
db.generate("B", n, 43, [n // 16, 300])
stmts = [
    "SELECT COUNT(*) FROM S, T;",
    "SELECT COUNT(*) FROM S, T WHERE vs < vt;",
    "SELECT COUNT(*) FROM S INNER JOIN T ON S.ks < T.kt;",
    "SELECT COUNT(*) FROM S INNER JOIN T ON S.ks = T.kt AND S.vs < T.vt;",
    "SELECT COUNT(*) FROM S INNER JOIN T ON S.ks = T.kt OR S.vs = T.vt;",
    "SELECT COUNT(*) FROM A WHERE fa IN (1, 2, 3, 500, 999) OR fa IS NULL;",
    "SELECT COUNT(*) FROM A WHERE fa < 10 XOR id_a < 1000000;",
    "SELECT COUNT(*) FROM A WHERE fa IS NOT NULL AND fa <> 5 AND fa >= 3 AND fa <= 900;",
    "SELECT COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b WHERE fa < fb;",
    "SELECT fa, fb, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY fa, fb LIMIT 5;",
]
for q in stmts:
    try:
        db.query(q)
        t0 = time.perf_counter()
        r = db.query(q)
        wall = (time.perf_counter() - t0) * 1e3
        first = r.rows()[0] if r.nrows else None
        print(f"{r.exec_ms:9.3f} ms exec {wall:9.3f} ms wall  rows {r.nrows:>8} first {first}  {q}")
    except Exception as e:
        print("ERROR", q, str(e)[:300])
db.close()
