"""GROUP BY over a primary key through query_execute() (10^8 rows, results kept on the device): the catalog's measured "no value twice"
makes it the identity - and since round 6 the stream stays the identity too (no row-id vector, no gather in the projection)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from midoridb_amd.query import DB
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
with DB() as db:
    db.execute("CREATE TABLE A (id_a INT, x INT);")
    db.generate_shard("A", n, 0, n, 42, [0, 0])
    db.results_on_device(True)
    for knob in ("0", "1", "0", "1"):
        os.environ["MDB_GROUP_IDENTITY"] = knob
        for _ in range(2):
            db.query_device("SELECT id_a, COUNT(*) FROM A GROUP BY id_a;", copy=False)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            r = db.query_device("SELECT id_a, COUNT(*) FROM A GROUP BY id_a;", copy=False)
        torch.cuda.synchronize()
        print(f"MDB_GROUP_IDENTITY={knob}: {(time.perf_counter() - t) / 5 * 1e3:.3f} ms per statement, {r[3]} groups, group_form {db.last_plan()['group_form']}", flush=True)
