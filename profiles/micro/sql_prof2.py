"""Per-kernel device time of statements on unique-key tables (MDB_PROF_DUMP=1)."""
import os, sys
os.environ["MDB_PROF_DUMP"] = "1"
sys.path.insert(0, '.')
from midoridb_amd.query import DB
n = 100_000_000
db = DB()
db.execute("CREATE TABLE A (id_a INT, x INT);")
db.execute("CREATE TABLE B (id_b INT, y INT);")
db.generate("A", n, 42, [0, 0])
db.generate("B", n, 43, [0, 0])
for q in sys.argv[1:]:
    db.query(q)
    db.query(q)
db.close()
