#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the REAL reference (oracle/_ref/libmidori_ref.so, built
from /root/reference by `make -C oracle ref`) on small inputs.  Run in the authoring container:

    python tests/golden/make_golden.py

A fixture is data only: the SQL, the input tables and the rows the reference returned
(column names in the reference's physical result order, values as query_column_int64() would
return them).  Every case stays inside the domain where the reference implements SQL semantics
(SURVEY.md 8a): GROUP BY / COUNT inputs fit one 4 KiB datablock (D1), no multi-value IN (D3),
at least one row reaches the early-materialisation table (D8), integers within int32 (D5).
The 3-way join cases are produced by chaining two reference 2-way joins through a real
intermediate table, because the reference's own recursive join is defective (D2).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def run_case(name, ddl, tables, query, source, via=None):
    """ddl: list of CREATE statements; tables: {name: (cols list of lists, nulls list of lists or None)}."""
    if os.environ.get("GOLDEN_TRACE"):
        print("case", name, query, flush=True)
    db = ref.RefDB()
    for s in ddl:
        db.execute(s)
    for t, (cols, nulls) in tables.items():
        if len(cols[0]):
            db.bulk_insert(t, [np.array(c) for c in cols], None if nulls is None else [None if x is None else np.array(x) for x in nulls])
    if via is not None:
        names, rows = via(db)
    else:
        names, rows = db.query(query)
    db.close()
    return {
        "name": name, "source": source, "ddl": ddl,
        "tables": {t: {"cols": [[(float(v) if isinstance(v, float) else int(v)) for v in c] for c in cols],
                       "nulls": None if nulls is None else [None if x is None else [int(b) for b in x] for x in nulls]}
                   for t, (cols, nulls) in tables.items()},
        "query": query, "expect": {"names": names, "rows": [list(r) for r in rows]},
    }


def reference_tests():
    """The reference's own known-answer cases (tests/engine/executor_select.c), same SQL, same data."""
    T = "reference tests/engine/executor_select.c"
    A3 = {"A": ([[1, 2, 3], [123, 456, 789]], None), "B": ([[1, 3], [-12345, -67890]], None)}
    ddl_ab = ["CREATE TABLE A (id_a INT, f1 INT);", "CREATE TABLE B (id_b INT, f2 INT);"]
    cases = [
        run_case("ref_select_1", ["CREATE TABLE TEST (f1 INT);"], {"TEST": ([[123, -12345]], None)}, "SELECT * FROM TEST;", T + ":47"),
        run_case("ref_select_2", ["CREATE TABLE A (f1 INT);", "CREATE TABLE B (f2 INT);"],
                 {"A": ([[123, 456]], None), "B": ([[-12345, -67890]], None)}, "SELECT * FROM A, B;", T + ":71"),
        run_case("ref_select_3", ddl_ab, A3, "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b;", T + ":102"),
        run_case("ref_select_5", ddl_ab, A3, "SELECT f1, f2 FROM A INNER JOIN B ON A.id_a = B.id_b;", T + ":168"),
        run_case("ref_select_6", ddl_ab, A3, "SELECT f1, f2 FROM A INNER JOIN B ON A.id_a = B.id_b WHERE f1 = 123;", T + ":197"),
        run_case("ref_select_7", ddl_ab, A3, "SELECT f1, f2 FROM A INNER JOIN B ON A.id_a = B.id_b WHERE 123 >= f1 AND f1 < 200;", T + ":231"),
        run_case("ref_select_9", ["CREATE TABLE A (id INT, f1 INT);"],
                 {"A": ([[1, 2, 3, 4, 5], [1, 2, 0, 4, 0]], [None, [0, 0, 1, 0, 1]])}, "SELECT id FROM A WHERE f1 IS NULL;", T + ":292"),
        run_case("ref_select_10", ["CREATE TABLE A (id INT, f1 INT);"],
                 {"A": ([[1, 1, 3, 3, 4], [1, 2, 0, 4, 0]], [None, [0, 0, 1, 0, 1]])}, "SELECT id, COUNT(*) FROM A GROUP BY id;", T + ":318"),
        run_case("ref_select_11", ["CREATE TABLE A (id_a INT);", "CREATE TABLE B (id_b INT);"],
                 {"A": ([[1, 3, 4]], None), "B": ([[1, 1, 3, 3, 4, 0]], [[0, 0, 0, 0, 0, 1]])},
                 "SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a;", T + ":348 (README example, north-star query)"),
        run_case("ref_select_12", ["CREATE TABLE A (id INT);"], {"A": ([[1, 3, 4]], None)}, "SELECT COUNT(*) FROM A WHERE id > 1;", T + ":380"),
        # case 8 with a single IN value (the multi-value form is reference defect D3)
        run_case("ref_select_8_single_in", ["CREATE TABLE A (f1 INT);"], {"A": ([[1, 2, 123, 3, 126, 4, 124, 125]], None)},
                 "SELECT f1 FROM A WHERE f1 IN (124);", T + ":265 (restricted to one IN value, D3)"),
    ]
    return cases


def three_way(db_tables, ddl, name, source):
    """(A JOIN B) JOIN C expected rows from two chained reference 2-way joins (D2 workaround)."""
    q = "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c;"

    def via(db):
        n1, r1 = db.query("SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b;")
        # materialise A JOIN B as a real table AB with the same bare column names, in (A cols, B cols) order
        bare = [n.split(".")[1] for n in n1]
        db.execute("CREATE TABLE AB (" + ", ".join(c + " INT" for c in bare) + ");")
        if r1:
            db.bulk_insert("AB", [np.array([r[k] for r in r1]) for k in range(len(bare))])
        n2, r2 = db.query("SELECT * FROM AB INNER JOIN C ON AB.id_a = C.id_c;")
        # rename AB.x back to the owning table, keep the values; the direct 3-way query's column ORDER is
        # checked separately against the djb2 emulation, so store names sorted into that order by the test
        owner = {c: "A" for c in db_tables["A_cols"]}
        owner.update({c: "B" for c in db_tables["B_cols"]})
        names = [(owner[n.split(".")[1]] + "." + n.split(".")[1]) if n.startswith("AB.") else n for n in n2]
        return names, r2

    return run_case(name, ddl, db_tables["data"], q, source, via=via)


def probes():
    S = "reference executor behind its parser seam (oracle/_ref), probe"
    ddl = ["CREATE TABLE A (id_a INT, f1 INT);", "CREATE TABLE B (id_b INT, f2 DOUBLE);"]
    f2 = np.array([0.5, 1.5, 2.5])
    data = {"A": ([[1, 3, 4, 3], [10, 30, 0, 31]], [None, [0, 0, 1, 0]]), "B": ([[1, 3, 3], f2.tolist()], None)}

    def rc(name, q):
        d = {"A": data["A"], "B": ([data["B"][0][0], f2], None)}
        c = run_case(name, ddl, d, q, S)
        return c
    qs = [
        ("probe_count_empty", "SELECT COUNT(*) FROM A WHERE id_a > 100;"),
        ("probe_where_empty", "SELECT id_a FROM A WHERE id_a > 100;"),
        ("probe_count_all", "SELECT COUNT(*) FROM A;"),
        ("probe_group_count_order", "SELECT f1, COUNT(*) FROM A GROUP BY f1;"),
        ("probe_count_first", "SELECT COUNT(*), id_a FROM A GROUP BY id_a;"),
        ("probe_double_where", "SELECT f2 FROM B WHERE f2 > 1.0;"),
        ("probe_join_double_where", "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b WHERE f2 < 2.0;"),
        ("probe_group_by_right_key", "SELECT id_b, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_b;"),
        ("probe_not_in", "SELECT f1 FROM A WHERE id_a NOT IN (3, 4);"),
        ("probe_eq_null", "SELECT id_a FROM A WHERE f1 = NULL;"),
        ("probe_const_true", "SELECT id_a FROM A WHERE 1 = 1;"),
        ("probe_or", "SELECT id_a FROM A WHERE id_a = 3 OR f1 = 10;"),
        ("probe_xor", "SELECT id_a FROM A WHERE id_a = 3 XOR f1 = 30;"),
        ("probe_qualified", "SELECT A.id_a FROM A WHERE A.id_a <> 3;"),
        ("probe_alias", "SELECT x.id_a FROM A AS x WHERE x.f1 >= 30;"),
        ("probe_on_residual", "SELECT id_a, f1 FROM A INNER JOIN B ON A.id_a = B.id_b AND f1 > 10;"),
        ("probe_is_not_null", "SELECT id_a FROM A WHERE f1 IS NOT NULL;"),
        ("probe_yoda", "SELECT f1 FROM A WHERE 10 < f1;"),
        ("probe_cross_where", "SELECT id_a, id_b FROM A, B WHERE id_a < id_b;"),
        ("probe_on_swapped", "SELECT id_a, id_b FROM A INNER JOIN B ON B.id_b = A.id_a;"),
    ]
    return [rc(n, q) for n, q in qs]


def randomized(seed, count):
    """Random in-domain cases: tables of <= 40 rows, keys from a small domain with NULLs."""
    rng = np.random.default_rng(seed)
    cases = []
    ddl = ["CREATE TABLE A (id_a INT, f1 INT);", "CREATE TABLE B (id_b INT, f2 INT);", "CREATE TABLE C (id_c INT, f3 INT);"]
    templates = [
        "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b;",
        "SELECT f1, f2 FROM A INNER JOIN B ON A.id_a = B.id_b WHERE f1 > {k} AND f2 <= {m};",
        "SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a;",
        "SELECT id_b, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_b;",
        "SELECT id_a, COUNT(*) FROM A GROUP BY id_a;",
        "SELECT f1, COUNT(*) FROM A GROUP BY f1;",
        "SELECT COUNT(*) FROM A WHERE f1 < {k} OR id_a = {m};",
        "SELECT id_a, f1 FROM A WHERE f1 IS NOT NULL;",
        "SELECT id_a FROM A WHERE f1 <> {k} XOR id_a < {m};",
        "SELECT id_a, id_b, f2 FROM A INNER JOIN B ON A.id_a = B.id_b AND f2 > {k};",
        "SELECT f1, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY f1;",
        "SELECT COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b WHERE f1 >= {k};",
    ]
    for i in range(count):
        na, nb = int(rng.integers(1, 9)), int(rng.integers(1, 9))	# <= 8 x 8 = 64 joined rows (one datablock, D1)
        dom = int(rng.integers(2, 6))
        a_id = rng.integers(0, dom, na)
        b_id = rng.integers(0, dom, nb)
        b_id[0] = a_id[0]						# at least one joined row (D8)
        a_f = rng.integers(-50, 50, na)
        b_f = rng.integers(-50, 50, nb)
        b_f[0] = 49							# keeps the guaranteed pair alive under "AND f2 > k" (D8)
        a_idn = (rng.random(na) < 0.15).astype(int)
        a_idn[0] = 0
        b_idn = (rng.random(nb) < 0.15).astype(int)
        b_idn[0] = 0
        a_fn = (rng.random(na) < 0.2).astype(int)
        tables = {"A": ([a_id.tolist(), a_f.tolist()], [a_idn.tolist(), a_fn.tolist()]),
                  "B": ([b_id.tolist(), b_f.tolist()], [b_idn.tolist(), None]),
                  "C": ([[], []], None)}
        t = templates[i % len(templates)]
        q = t.format(k=int(rng.integers(-30, 30)), m=int(rng.integers(0, dom)))
        if "WHERE" in q and "JOIN" not in q and "COUNT" in q:
            pass
        try:
            cases.append(run_case(f"random_{seed}_{i}", ddl, tables, q, "randomised in-domain case run through oracle/_ref"))
        except ref.RefError as e:
            print("skip", q, e)
    return cases


def three_way_cases(seed, count):
    rng = np.random.default_rng(seed)
    ddl = ["CREATE TABLE A (id_a INT, f1 INT);", "CREATE TABLE B (id_b INT, f2 INT);", "CREATE TABLE C (id_c INT, f3 INT);"]
    out = []
    # the reference's own test data for the 3-way join (tests/engine/executor_select.c:133-166)
    fixed = {"A": ([[1, 2, 3], [123, 456, 789]], None), "B": ([[1, 2, 3], [-12345, -11111, -67890]], None),
             "C": ([[1, 3, 4], [333, 666, 999]], None)}
    out.append(three_way({"A_cols": ["id_a", "f1"], "B_cols": ["id_b", "f2"], "data": fixed}, ddl, "ref_select_4_intended",
                         "reference tests/engine/executor_select.c:133 (expected rows of the test; the reference itself returns only "
                         "the first, defect D2) - produced by chaining two reference 2-way joins"))
    for i in range(count):
        na, nb, nc = (int(rng.integers(2, 8)) for _ in range(3))
        dom = int(rng.integers(2, 5))
        a, b, c = rng.integers(0, dom, na), rng.integers(0, dom, nb), rng.integers(0, dom, nc)
        b[0] = a[0]
        c[0] = a[0]
        data = {"A": ([a.tolist(), rng.integers(0, 99, na).tolist()], None), "B": ([b.tolist(), rng.integers(0, 99, nb).tolist()], None),
                "C": ([c.tolist(), rng.integers(0, 99, nc).tolist()], None)}
        out.append(three_way({"A_cols": ["id_a", "f1"], "B_cols": ["id_b", "f2"], "data": data}, ddl, f"three_way_{seed}_{i}",
                             "chained reference 2-way joins (D2 workaround)"))
    return out


def column_orders():
    """Result column order (R3) of SELECT * over assorted schemas, straight from the reference."""
    out = []
    schemas = [
        (["CREATE TABLE A (id_a INT, f1 INT);", "CREATE TABLE B (id_b INT, f2 INT);"], "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b;"),
        (["CREATE TABLE T1 (a INT, b INT, c INT, d INT, e INT);", "CREATE TABLE T2 (f INT, g INT, h INT, i INT);"],
         "SELECT * FROM T1 INNER JOIN T2 ON T1.a = T2.f;"),
        (["CREATE TABLE orders (order_id INT, customer INT, amount INT, region INT, status INT, flag INT);",
          "CREATE TABLE cust (cust_id INT, segment INT, country INT);"], "SELECT * FROM orders INNER JOIN cust ON orders.customer = cust.cust_id;"),
        (["CREATE TABLE A (id_a INT, f1 INT);", "CREATE TABLE B (id_b INT, f2 INT);", "CREATE TABLE C (id_c INT, f3 INT);"],
         "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c;"),
        (["CREATE TABLE W (c1 INT, c2 INT, c3 INT, c4 INT, c5 INT, c6 INT, c7 INT, c8 INT, c9 INT, c10 INT, c11 INT, c12 INT);"], "SELECT * FROM W;"),
    ]
    for k, (ddl, q) in enumerate(schemas):
        db = ref.RefDB()
        tabs = {}
        for s in ddl:
            db.execute(s)
            name = s.split()[2]
            ncols = s.count(" INT")
            db.bulk_insert(name, [np.array([7])] * ncols)
            tabs[name] = ([[7]] * ncols, None)
        names, rows = db.query(q)
        db.close()
        out.append({"name": f"column_order_{k}", "source": "reference result column order (djb2 hashtable iteration, SURVEY 8a R3)",
                    "ddl": ddl, "tables": {t: {"cols": c, "nulls": None} for t, (c, _) in tabs.items()}, "query": q,
                    "expect": {"names": names, "rows": [list(r) for r in rows]}})
    return out


def config1():
    """BASELINE.json configs[0]: single-table SELECT + WHERE over 1M INT64 rows through the reference
    executor (CPU).  Too large to store row by row: the fixture keeps the row count, the first and last
    rows and a checksum (sum and xor of all returned values)."""
    n = 1_000_000
    db = ref.RefDB()
    db.execute("CREATE TABLE T (v INT);")
    db.bulk_insert("T", [np.arange(n, dtype=np.int64)])
    q = "SELECT v FROM T WHERE v > 500000;"
    db.execute(q)
    names, vals, _ = db.fetch()
    db.close()
    col = vals[:, 0]
    return [{"name": "config1_scan_where_1M", "source": "BASELINE.json configs[0] via oracle/_ref (reference CPU executor)",
             "ddl": ["CREATE TABLE T (v INT);"], "generator": {"table": "T", "column": "v", "rows": n, "values": "v = i"}, "query": q,
             "expect": {"names": names, "nrows": int(len(col)), "first": int(col[0]), "last": int(col[-1]), "sum": int(col.sum()),
                        "xor": int(np.bitwise_xor.reduce(col))}}]


def run_script(name, source, stmts, table_names):
    """A DML script through the real reference: every statement's status, rows affected and, after it, the live
    rows of every table in scan order (NULL cells reported as 0 + flag: the bytes under a NULL are not observable
    through the public API)."""
    db = ref.RefDB()
    steps = []
    created = set()
    for sql in stmts:
        st = {"sql": sql}
        if os.environ.get("GOLDEN_TRACE"):
            print("script", name, sql, flush=True)
        if sql.upper().startswith("SELECT") and any(db.table_rows(t) == 0 for t in created):
            continue	# the reference crashes when a SELECT scans a table without live rows
        try:
            rc = db.execute(sql)
            st["status"] = "ok"
            if rc == 1:
                names, rows = db.query(sql)
                st["result"] = {"names": names, "rows": [list(r) for r in rows]}
            else:
                st["rows_affected"] = db.rows_affected()
        except ref.RefError as e:
            st["status"] = "error"
            st["error"] = str(e)
        for t in table_names:
            if sql.upper().startswith("CREATE TABLE " + t.upper() + " "):
                created.add(t)
        if not sql.upper().startswith(("INSERT", "CREATE")) or sql is stmts[-1]:
            st["tables"] = {}
            for t in sorted(created):		# live rows in scan order, column-major, null = NULL cell
                vals, nulls = db.table_dump(t)
                st["tables"][t] = [[None if nulls[r, c] else int(vals[r, c]) for r in range(vals.shape[0])] for c in range(vals.shape[1])]
        steps.append(st)
    db.close()
    return {"name": name, "source": source, "steps": steps}


def dml_cases(seed, count):
    """DELETE / UPDATE: the reference's own known-answer statements (tests/engine/executor_delete.c,
    executor_update.c; INT and DOUBLE columns) plus randomised scripts mixing INSERT, DELETE, UPDATE and SELECT
    on a three-column table with NULLs that spans several 4 KiB datablocks."""
    TD, TU = "reference tests/engine/executor_delete.c", "reference tests/engine/executor_update.c"
    ints = [123, 456, 789, 101112, -789, -12345]
    cases = []
    ops = ["=", ">", ">=", "<", "<=", "<>"]
    for typ, lit, fmt in (("INT", "123", lambda v: str(v)), ("DOUBLE", "123.0", lambda v: f"{v}.0")):
        base = [f"CREATE TABLE TEST (f1 {typ});"] + [f"INSERT INTO TEST VALUES ({fmt(v)});" for v in ints]
        for i, op in enumerate(ops):
            cases.append(run_script(f"ref_delete_{typ.lower()}_{i}", TD, base + [f"DELETE FROM TEST WHERE f1 {op} {lit};", "SELECT * FROM TEST;"], ["TEST"]))
            cases.append(run_script(f"ref_update_{typ.lower()}_{i}", TU,
                                    base + [f"UPDATE TEST SET f1 = {'42' if typ == 'INT' else '42.0'} WHERE f1 {op} {lit};", "SELECT * FROM TEST;"], ["TEST"]))
    base = ["CREATE TABLE TEST (f1 INT);"] + [f"INSERT INTO TEST VALUES ({v});" for v in (123, 456, -789)] + ["INSERT INTO TEST VALUES (NULL);"]
    cases.append(run_script("ref_delete_all", TD + ":99", base + ["DELETE FROM TEST;", "INSERT INTO TEST VALUES (7);", "SELECT * FROM TEST;"], ["TEST"]))
    cases.append(run_script("ref_delete_null_cmp", TD + ":1039-1044",
                            base + ["DELETE FROM TEST WHERE f1 = NULL;", "DELETE FROM TEST WHERE f1 != NULL;", "DELETE FROM TEST WHERE f1 > NULL;",
                                    "DELETE FROM TEST WHERE f1 >= NULL;", "DELETE FROM TEST WHERE f1 < NULL;", "DELETE FROM TEST WHERE f1 <= NULL;"], ["TEST"]))
    cases.append(run_script("ref_delete_is_null", TD + ":1055", base + ["DELETE FROM TEST WHERE f1 IS NULL;", "SELECT * FROM TEST;"], ["TEST"]))
    cases.append(run_script("ref_delete_is_not_null", TD + ":1066", base + ["DELETE FROM TEST WHERE f1 IS NOT NULL;", "SELECT * FROM TEST;"], ["TEST"]))
    cases.append(run_script("ref_update_all", TU, base + ["UPDATE TEST SET f1=42;", "SELECT * FROM TEST;"], ["TEST"]))
    cases.append(run_script("ref_update_null_cmp", TU,
                            base + ["UPDATE TEST SET f1 = 42 WHERE f1 = NULL;", "UPDATE TEST SET f1 = 42 WHERE f1 != NULL;", "UPDATE TEST SET f1 = 42 WHERE f1 > NULL;",
                                    "UPDATE TEST SET f1 = 42 WHERE f1 <= NULL;", "UPDATE TEST SET f1 = 42 WHERE f1 IS NULL;", "SELECT * FROM TEST;"], ["TEST"]))
    g = ["CREATE TABLE G (f1 INT, f2 INT);", "INSERT INTO G VALUES (123, 123);", "INSERT INTO G VALUES (456, 123);", "INSERT INTO G VALUES (789, 987);",
         "INSERT INTO G VALUES (101112, NULL);"]
    for i, (src, tail) in enumerate([
            (TU + ":2147", ["UPDATE G SET f1=42, f2=43WHERE 1 = 1;"]),
            (TU, ["UPDATE G SET f1=42, f2=43 WHERE f1 = f2;"]),
            (TU, ["UPDATE G SET f1=42, f2=43 WHERE f1 > f2;"]),
            (TU, ["UPDATE G SET f1=42, f2=43 WHERE f1 IN (456, 789) AND f2 NOT IN (123);"]),
            (TU + ":2215", ["UPDATE G SET f1=42, f2=43 WHERE (f2 < 1000 AND f2 > 100) XOR (f1 > 100 OR f1 > 10000);"]),
            (TD + ":1720", ["DELETE FROM G WHERE f1 = NULL;"]),
            (TD + ":1897", ["DELETE FROM G WHERE 1 = NULL;"]),
            (TD, ["DELETE FROM G WHERE f1 = f2;"]),
            (TD, ["DELETE FROM G WHERE f1 > f2 OR f2 IS NULL;"]),
            (TD, ["DELETE FROM G WHERE f1 IN (456, 789) AND f2 NOT IN (123);"]),
            ("probe: SET NULL then IS NULL, INSERT after DELETE keeps scan order",
             ["UPDATE G SET f2 = NULL WHERE f1 < 500;", "DELETE FROM G WHERE f1 = 456;", "INSERT INTO G VALUES (5, 6);", "UPDATE G SET f2 = 1 WHERE f2 IS NULL;",
              "SELECT f1 FROM G WHERE f2 = 1;"]),
            ("probe: errors leave the table untouched",
             ["DELETE FROM G WHERE nosuch = 1;", "UPDATE G SET nosuch = 1;", "UPDATE G SET f1 = 1.5;", "DELETE FROM G WHERE f1 = 1.5;", "DELETE FROM NOSUCH;",
              "UPDATE G SET f1 = 3 WHERE f2 < NULL;"])]):
        cases.append(run_script(f"ref_multi_{i}", src, g + tail + ["SELECT * FROM G;"], ["G"]))

    # ---- randomised scripts
    rng = np.random.default_rng(seed)

    def lit_int():
        return str(int(rng.integers(0, 12)))

    def pred():
        k = int(rng.integers(0, 9))
        c = ["a", "b"][int(rng.integers(0, 2))]
        op = ops[int(rng.integers(0, 6))]
        if k == 0:
            return f"{c} {op} {lit_int()}"
        if k == 1:
            return f"a {op} b"
        if k == 2:
            return f"{c} IS {'NOT ' if rng.integers(0, 2) else ''}NULL"
        if k == 3:
            return f"{c} IN ({', '.join(lit_int() for _ in range(int(rng.integers(1, 4))))})"
        if k == 4:
            return f"{c} NOT IN ({lit_int()})"
        if k == 5:
            return f"x {op} {rng.integers(0, 8) / 8.0 + 0.0:.3f}"
        if k == 6:
            return f"({c} {op} {lit_int()} AND a {ops[int(rng.integers(0, 6))]} {lit_int()}) OR x IS NULL"
        if k == 7:
            return f"{c} {op} {lit_int()} XOR b {ops[int(rng.integers(0, 6))]} {lit_int()}"
        return f"{c} = {lit_int()} OR ({c} > {lit_int()} AND x < 0.5)"

    for i in range(count):
        n = int(rng.choice([5, 40, 90, 200]))
        stmts = ["CREATE TABLE T (a INT, b INT, x DOUBLE);"]
        for _ in range(n):
            a = "NULL" if rng.random() < 0.1 else lit_int()
            b = "NULL" if rng.random() < 0.1 else lit_int()
            x = "NULL" if rng.random() < 0.1 else f"{rng.integers(0, 8) / 8.0:.3f}"
            stmts.append(f"INSERT INTO T VALUES ({a}, {b}, {x});")
        for _ in range(int(rng.integers(3, 8))):
            k = int(rng.integers(0, 10))
            if k < 4:
                stmts.append(f"DELETE FROM T WHERE {pred()};")
            elif k < 8:
                sets = []
                for c in rng.permutation(["a", "b", "x"])[: int(rng.integers(1, 3))]:
                    if rng.random() < 0.2:
                        sets.append(f"{c} = NULL")
                    elif c == "x":
                        sets.append(f"x = {rng.integers(0, 8) / 8.0:.3f}")
                    else:
                        sets.append(f"{c} = {lit_int()}")
                stmts.append(f"UPDATE T SET {', '.join(sets)}" + (f" WHERE {pred()};" if rng.random() < 0.85 else ";"))
            elif k == 8:
                stmts.append(f"INSERT INTO T VALUES ({lit_int()}, {lit_int()}, 0.250);")
            else:
                p = pred()
                while (" IS " in p and any(w in p for w in (" AND ", " OR ", " XOR "))) or (" IN (" in p and "," in p):
                    p = pred()	# SELECT only: IS NULL under AND/OR is rejected upstream, multi-value IN is defect D3
                stmts.append(f"SELECT a, b FROM T WHERE {p};")
        stmts.append("SELECT * FROM T;")
        cases.append(run_script(f"dml_random_{i}", "randomised, via oracle/_ref", stmts, ["T"]))
    return cases


def double_joins(seed, count):
    """Equi-joins on DOUBLE keys: the reference compares them with IEEE `==` (cmp_double_value_to_value,
    executor_select.c:440-460), so -0.0 joins +0.0 and NaN joins nothing - the two places where comparing the
    8-byte words would differ.  NaN cannot be written as a literal: the rows enter through the bulk insert."""
    S = "reference executor via oracle/_ref: DOUBLE join keys (+-0.0, NaN, duplicates, NULLs)"
    ddl = ["CREATE TABLE A (x DOUBLE, fa INT);", "CREATE TABLE B (y DOUBLE, fb INT);"]
    nan = float("nan")
    fixed = {"A": ([[0.0, -0.0, nan, 1.5, 1.5, 2.0, 7.25], [1, 2, 3, 4, 5, 6, 7]], [[0, 0, 0, 0, 0, 0, 1], None]),
             "B": ([[-0.0, 0.0, nan, 1.5, 3.0, 7.25], [10, 20, 30, 40, 50, 60]], [[0, 0, 0, 0, 0, 1], None])}
    qs = ["SELECT * FROM A INNER JOIN B ON A.x = B.y;",
          "SELECT fa, fb FROM A INNER JOIN B ON B.y = A.x WHERE fa > 1;",
          "SELECT fa, COUNT(*) FROM A INNER JOIN B ON A.x = B.y GROUP BY fa;",
          "SELECT COUNT(*) FROM A INNER JOIN B ON A.x = B.y;",
          "SELECT fa, fb FROM A INNER JOIN B ON A.x = B.y AND fb > 10;"]
    cases = [run_case(f"double_join_fixed_{i}", ddl, fixed, q, S) for i, q in enumerate(qs)]
    rng = np.random.default_rng(seed)
    pool = np.array([0.0, -0.0, nan, 0.5, 1.5, -1.5, 1e300, -1e-300])
    for i in range(count):
        na, nb = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        x, y = pool[rng.integers(0, len(pool), na)], pool[rng.integers(0, len(pool), nb)]
        x[0] = 0.5
        y[0] = 0.5			# at least one joined row (D8)
        xn, yn = (rng.random(na) < 0.15).astype(int), (rng.random(nb) < 0.15).astype(int)
        xn[0] = yn[0] = 0
        tables = {"A": ([x.tolist(), rng.integers(0, 5, na).tolist()], [xn.tolist(), None]),
                  "B": ([y.tolist(), rng.integers(0, 5, nb).tolist()], [yn.tolist(), None])}
        cases.append(run_case(f"double_join_random_{seed}_{i}", ddl, tables, qs[i % 4], S))
    return cases


def typed_scripts():
    """Tables that hold every reference column type (include/primitive/column.h:17-25): statements that reference the
    INT / DOUBLE / DATE / DATETIME / TINYINT columns of tables which also contain VARCHAR columns, through the real
    reference.  Recorded per statement: status, rows affected, SELECT results (a DATE / DATETIME cell reads as its time_t,
    a TINYINT cell as 0 | 1); generated with TZ=UTC (mktime)."""
    os.environ["TZ"] = "UTC"
    import time
    time.tzset()
    S = "reference executor via oracle/_ref: mixed column types"
    setup = [
        "CREATE TABLE P (id INT, name VARCHAR(12), born DATE, seen DATETIME, ok TINYINT, w DOUBLE);",
        "INSERT INTO P VALUES (1, 'ann', '1990-05-17', '2023-06-02 10:11:12', TRUE, 0.5);",
        "INSERT INTO P VALUES (2, 'bob', '2001-02-03', '2023-06-03 00:00:00', FALSE, 1.5);",
        "INSERT INTO P VALUES (3, NULL, NULL, NULL, NULL, NULL);",
        "INSERT INTO P VALUES (4, 'dee', '2001-02-03', '2024-01-01 23:59:59', TRUE, 2.5);",
        "INSERT INTO P (id, born) VALUES (5, '1970-01-02');",
        "CREATE TABLE Q (qid INT, day DATE, note VARCHAR(5));",
        "INSERT INTO Q VALUES (10, '2001-02-03', 'x'), (11, '1990-05-17', NULL), (12, '2030-12-31', 'zz'), (13, NULL, 'n');",
    ]
    scripts = {
        "typed_select": setup + [
            "SELECT id, born FROM P WHERE id > 1;",
            "SELECT id, seen, ok FROM P WHERE born IS NOT NULL;",
            "SELECT id FROM P WHERE ok = TRUE;",
            "SELECT id FROM P WHERE ok <> TRUE;",
            "SELECT id, w FROM P WHERE w >= 1.5 AND ok = FALSE;",
            "SELECT id, qid FROM P INNER JOIN Q ON P.born = Q.day;",
            "SELECT born, COUNT(*) FROM P INNER JOIN Q ON P.born = Q.day GROUP BY born;",
            "SELECT COUNT(*) FROM P WHERE seen IS NULL;",
            "SELECT id FROM P WHERE born = '2001-02-03';",
            "SELECT id FROM P WHERE id = 'abc';",
            # VARCHAR: equality only; cells travel as ids of the database's string dictionary on the device
            "SELECT id, name FROM P;",
            "SELECT id FROM P WHERE name = 'bob';",
            "SELECT id FROM P WHERE name <> 'bob';",
            "SELECT id FROM P WHERE name = 'nobody';",
            "SELECT id FROM P WHERE name <> 'nobody';",
            "SELECT id, name FROM P WHERE name IS NULL;",
            "SELECT id FROM P WHERE name > 'ann';",
            "SELECT id FROM P WHERE name = 5;",
            "SELECT qid, note FROM Q WHERE note = 'zz' OR note = 'x';",
        ],
        "typed_varchar_join_group": [
            "CREATE TABLE C (cid INT, city VARCHAR(8));",
            "CREATE TABLE V (vid INT, town VARCHAR(8));",
            "INSERT INTO C VALUES (1, 'oslo'), (2, 'rome'), (3, 'oslo'), (4, NULL), (5, 'bern'), (6, 'rome'), (7, 'oslo');",
            "INSERT INTO V VALUES (10, 'rome'), (11, 'oslo'), (12, 'paris'), (13, NULL), (14, 'oslo');",
            "SELECT cid, vid FROM C INNER JOIN V ON C.city = V.town;",
            "SELECT city, COUNT(*) FROM C GROUP BY city;",
            "SELECT city, COUNT(*) FROM C INNER JOIN V ON C.city = V.town GROUP BY city;",
            "SELECT cid FROM C WHERE city IN ('rome');",
            "SELECT cid FROM C WHERE city NOT IN ('rome');",
        ],
        "typed_dml": setup + [
            "DELETE FROM P WHERE born < '2000-01-01';",
            "SELECT id, born FROM P;",
            "DELETE FROM Q WHERE note = 'x';",
            "SELECT qid, note FROM Q;",
            "DELETE FROM Q WHERE note <> 'zz';",
            "SELECT qid, note FROM Q;",
            "DELETE FROM Q WHERE note >= 'a';",
            "UPDATE Q SET note = 'new' WHERE note = 'zz';",
            "UPDATE Q SET note = 'toolong' WHERE qid = 11;",
            "SELECT qid, note FROM Q;",
            "UPDATE P SET seen = '2025-02-03 04:05:06', ok = FALSE WHERE id = 2;",
            "UPDATE P SET name = 'zed' WHERE id = 4;",
            "UPDATE P SET name = 'bob2' WHERE name = 'bob';",
            "SELECT id, name FROM P;",
            "UPDATE P SET born = NULL WHERE ok = TRUE;",
            "SELECT id, born, seen, ok FROM P;",
            "DELETE FROM P WHERE seen >= '2025-01-01 00:00:00';",
            "SELECT id FROM P;",
            "INSERT INTO P VALUES (7, 'way too long a name', NULL, NULL, NULL, NULL);",
            "INSERT INTO P VALUES (7, 'ok', '2020-13-45', NULL, NULL, NULL);",
            "INSERT INTO P VALUES (7, 'ok', NULL, NULL, 5, NULL);",
            "SELECT id FROM P;",
        ],
        # the reference's own DELETE / UPDATE tests over DATE and VARCHAR columns (tests/engine/executor_delete.c:1107-1560,
        # tests/engine/executor_update.c:1420-1975): the same tables and statements, the table re-read after each one (plus a
        # NULL row, which no comparison matches: it keeps the table non-empty - the reference crashes on a SELECT over a table
        # without live rows)
        "ref_delete_date": sum([[f"CREATE TABLE D{i} (f1 DATE);",
                                 f"INSERT INTO D{i} VALUES ('1990-01-01'), ('1991-01-01'), ('1992-01-01'), ('1993-01-01'), (NULL);",
                                 f"DELETE FROM D{i} WHERE f1 {op} '{lit}';", f"SELECT f1 FROM D{i};"]
                                for i, (op, lit) in enumerate([("=", "1990-01-01"), (">", "1990-01-01"), (">=", "1990-01-01"), ("<", "1991-01-01"),
                                                               ("<=", "1992-01-01"), ("<>", "1992-01-01")])], []),
        "ref_update_date": sum([[f"CREATE TABLE U{i} (f1 DATE);",
                                 f"INSERT INTO U{i} VALUES ('1990-01-01'), ('1991-01-01'), ('1992-01-01'), ('1993-01-01'), (NULL);",
                                 f"UPDATE U{i} SET f1 = '1993-01-01' WHERE f1 {op} '{lit}';", f"SELECT f1 FROM U{i};"]
                                for i, (op, lit) in enumerate([("=", "1990-01-01"), (">", "1990-01-01"), (">=", "1990-01-01"), ("<", "1991-01-01"),
                                                               ("<=", "1992-01-01"), ("<>", "1992-01-01")])], []),
        "ref_dml_varchar": [
            "CREATE TABLE TEST (f1 VARCHAR(4));",
            "INSERT INTO TEST VALUES ('123');", "INSERT INTO TEST VALUES ('456');", "INSERT INTO TEST VALUES (NULL);", "INSERT INTO TEST VALUES ('789');",
            "DELETE FROM TEST WHERE f1 > '123';", "DELETE FROM TEST WHERE f1 >= '456';", "DELETE FROM TEST WHERE f1 < NULL;", "DELETE FROM TEST WHERE f1 <= '789';",
            "UPDATE TEST SET f1='852' WHERE f1 > '123';", "UPDATE TEST SET f1='852' WHERE f1 >= '456';", "UPDATE TEST SET f1='852' WHERE f1 < NULL;",
            "UPDATE TEST SET f1='852' WHERE f1 <= '789';",
            "SELECT f1 FROM TEST;",
            "UPDATE TEST SET f1='852' WHERE f1 = '123';",
            "SELECT f1 FROM TEST;",
            "DELETE FROM TEST WHERE f1 = '852';",
            "SELECT f1 FROM TEST;",
            "DELETE FROM TEST WHERE f1 <> '456';",
            "SELECT f1 FROM TEST;",
            "UPDATE TEST SET f1 = NULL WHERE f1 = '456';",
            "SELECT f1 FROM TEST;",
        ],
        "typed_not_null": [
            "CREATE TABLE N (k INT PRIMARY KEY, v DOUBLE NOT NULL, note VARCHAR(8));",
            "INSERT INTO N VALUES (1, 0.5, 'a');",
            "INSERT INTO N VALUES (NULL, 0.5, 'a');",
            "INSERT INTO N VALUES (2, NULL, 'a');",
            "INSERT INTO N (k, note) VALUES (3, 'b');",
            "INSERT INTO N (v, k) VALUES (1.5, 4);",
            "SELECT k, v FROM N;",
        ],
    }
    out = []
    for name, stmts in scripts.items():
        db = ref.RefDB()
        steps = []
        for sql in stmts:
            st = {"sql": sql}
            try:
                rc = db.execute(sql)
                st["status"] = "ok"
                if rc == 1:
                    names, rows = db.query(sql)
                    st["result"] = {"names": names, "rows": [list(r) for r in rows]}
                else:
                    st["rows_affected"] = db.rows_affected()
            except (ref.RefError, ValueError) as e:
                st["status"] = "error"
                st["error"] = str(e)
            steps.append(st)
        db.close()
        out.append({"name": name, "source": S, "steps": steps})
    return out


def main():
    if not ref.available():
        sys.exit("oracle/_ref/libmidori_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    sets = {
        "reference_tests.json": reference_tests,
        "probes.json": probes,
        "randomized.json": lambda: randomized(20261002, 72),
        "three_way.json": lambda: three_way_cases(7, 8),
        "column_order.json": column_orders,
        "config1.json": config1,
        "dml.json": lambda: dml_cases(99, 30),
        "double_join.json": lambda: double_joins(5, 16),
        "typed_tables.json": typed_scripts,
    }
    only = set(sys.argv[1:])		# optional: file names to (re)generate; default all
    for fn, make in sets.items():
        if only and fn not in only:
            continue
        cases = make()
        with open(os.path.join(OUT, fn), "w") as f:
            json.dump(cases, f, indent=1)
        print(fn, len(cases), "cases")


if __name__ == "__main__":
    main()
