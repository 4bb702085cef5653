"""Latency of small queries through query_execute() (README example and friends): wall time per call."""
import sys, time
sys.path.insert(0, '.')
from midoridb_amd.query import DB
with DB() as db:
    db.execute("CREATE TABLE A (id_a INT);")
    db.execute("CREATE TABLE B (id_b INT);")
    db.execute("INSERT INTO A VALUES (1), (3), (4);")
    db.execute("INSERT INTO B VALUES (1), (1), (3), (3), (4), (NULL);")
    qs = ["SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a;",
          "SELECT * FROM A WHERE id_a > 1;",
          "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b;",
          "SELECT id_b, COUNT(*) FROM B GROUP BY id_b ORDER BY id_b DESC LIMIT 2;",
          "UPDATE A SET id_a = 3 WHERE id_a = 3;"]
    for q in qs:
        f = db.execute if q.startswith("UPDATE") else db.query
        for _ in range(20):
            f(q)
        t0 = time.perf_counter()
        for _ in range(200):
            f(q)
        print(f"{(time.perf_counter() - t0) / 200 * 1e6:8.1f} us  {q}")
