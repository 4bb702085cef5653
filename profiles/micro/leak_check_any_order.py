"""Repeated any-order queries (mdb_database_groups_any_order) through query_execute(): device and host memory must stay flat."""
import sys, os, torch, resource
sys.path.insert(0, '.')
import numpy as np
from midoridb_amd.query import DB
n = 3_000_000
with DB() as db:
    db.execute("CREATE TABLE A (id_a INT);")
    db.execute("CREATE TABLE B (id_b INT);")
    db.execute("CREATE TABLE C (id_c INT);")
    db.generate("A", n, 42, [0])
    db.generate("B", n, 43, [n // 16])
    db.generate("C", n, 44, [n // 8])
    db.groups_any_order(True)
    db.results_on_device(True)
    qs = ["SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a;",
          "SELECT COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b;",
          "SELECT id_b, COUNT(*) FROM B GROUP BY id_b;",
          "SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c GROUP BY id_a;"]
    def rss():
        return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024
    for it in range(5):
        for _ in range(30):
            for q in qs:
                db.query_device(q, copy=False)
        free, total = torch.cuda.mem_get_info()
        print(f"round {it}: device used {(total - free) / 2**20:8.0f} MiB   host max-rss {rss()} MiB", flush=True)
