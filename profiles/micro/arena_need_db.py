"""Scratch-arena size and wall time of the headline statement through query_execute() on a fresh database (tables generated on the device):
    python profiles/micro/arena_need_db.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from midoridb_amd.query import DB
from midoridb_amd.dev import _bind as _bind_dev
n = 100_000_000
SQL = "SELECT id_a, COUNT(*) FROM A INNER JOIN B ON id_a = id_b GROUP BY id_a;"
with DB() as db:
    db.execute("CREATE TABLE A (id_a INT);")
    db.execute("CREATE TABLE B (id_b INT);")
    _bind_dev(db.lib)
    db.lib.mdb_dev_arena_bytes.restype = ctypes.c_size_t
    db.lib.mdb_dev_arena_bytes.argtypes = [ctypes.c_void_p]
    if len(sys.argv) > 1:
        db.lib.mdb_dev_reserve(db.device_handle(), int(float(sys.argv[1])) << 30)
    db.generate("A", n, 42)
    db.generate("B", n, 43, [n // 16])
    db.results_on_device(True)
    for i in range(3):
        t0 = time.perf_counter()
        db.query_device(SQL, copy=False)
        print("call %d: %.2f ms, arena %.2f GB" % (i, (time.perf_counter() - t0) * 1e3, db.lib.mdb_dev_arena_bytes(db.device_handle()) / 1e9))
