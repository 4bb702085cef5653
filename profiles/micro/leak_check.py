"""Repeated mixed queries through query_execute(): device and host memory must stay flat after warm-up."""
import sys, os, torch, resource
sys.path.insert(0, '.')
import numpy as np
from midoridb_amd.query import DB
rng = np.random.default_rng(0)
n = 300_000
with DB() as db:
    db.execute("CREATE TABLE A (id_a INT, fa INT, xa DOUBLE);")
    db.execute("CREATE TABLE B (id_b INT, fb INT);")
    db.append_columns("A", [rng.integers(0, n // 4, n), rng.integers(0, 100, n), rng.random(n)], None)
    db.append_columns("B", [rng.integers(0, n // 4, n), rng.integers(0, 100, n)], None)
    qs = ["SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a;",
          "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b WHERE fa > 50 ORDER BY fb DESC, id_a LIMIT 100;",
          "SELECT fa, fb, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY fa, fb HAVING COUNT(*) > 3;",
          "SELECT DISTINCT fa FROM A WHERE xa < 0.5;",
          "UPDATE A SET fa = 7 WHERE fa = 8;",
          "DELETE FROM B WHERE fb = 99 AND id_b < 10;",
          "INSERT INTO B VALUES (1, 99);"]
    def rss():
        return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024
    for it in range(6):
        for _ in range(40):
            for q in qs:
                (db.query if q.startswith("SELECT") else db.execute)(q)
        free, total = torch.cuda.mem_get_info()
        print(f"round {it}: device used {(total - free) / 2**20:8.0f} MiB   host max-rss {rss()} MiB", flush=True)
