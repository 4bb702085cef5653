"""Per-kernel device time of single large statements (MDB_PROF_DUMP=1 makes query_execute() print it on stderr)."""
import os, sys
os.environ["MDB_PROF_DUMP"] = "1"
sys.path.insert(0, '.')
from midoridb_amd.query import DB
n = 100_000_000
db = DB()
db.execute("CREATE TABLE A (id_a INT, fa INT);")
db.execute("CREATE TABLE B (id_b INT, fb INT);")
db.generate("A", n, 42, [0, 1000])
db.generate("B", n, 43, [n // 16, 300])
for q in sys.argv[1:]:
    db.query(q)
    db.query(q)
db.close()
