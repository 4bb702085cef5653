"""Hand-written RPN token queues (the vocabulary of the reference grammar's emit(), reference
src/parser/midorisql.y:154-300) for every SELECT of the golden fixtures.

Why: the fixtures were produced by feeding the real reference the RPN that THIS repository's SQL front end
(midoridb_amd/csrc/mdb_sql.c) emitted, and the product executes the same RPN - a mis-translation into another valid query
would be invisible to parity.  These queues were written by hand from the grammar's productions, not generated:
tests/test_cpu_frontend_abi.py checks that mdb_sql.c produces exactly them, which ties every fixture to what the
reference's own bison parser would have handed its AST builder.

    select_stmt: SELECT opts exprs FROM refs [WHERE e] [GROUP BY ...] ... -> "SELECT <opts> <n>", n = exprs + top-level
                 table references (a join counts once) + clauses present;  "STMT" ends the statement
    join_table : <left ref> TABLE <right> <ON expr> ONEXPR JOIN 1;  table_factor: TABLE t [ALIAS a]
    expr       : postfix - operands, then CMP <1 < | 2 > | 3 <> | 4 = | 5 <= | 6 >=> / AND / OR / XOR / ISNULL / ISNOTNULL /
                 ISIN k / ISNOTIN k;  literals NUMBER n (signed), FLOAT %g, BOOL 0|1, STRING 'text', NULL
"""
import re

NORTH_AB = ["TABLE A", "TABLE B", "FIELDNAME A.id_a", "FIELDNAME B.id_b", "CMP 4", "ONEXPR", "JOIN 1"]
XY = ["TABLE A", "TABLE B", "FIELDNAME A.x", "FIELDNAME B.y", "CMP 4", "ONEXPR", "JOIN 1"]
PQ = ["TABLE P", "TABLE Q", "FIELDNAME P.born", "FIELDNAME Q.day", "CMP 4", "ONEXPR", "JOIN 1"]

FIXED = {
    # reference tests/engine/executor_select.c
    "SELECT * FROM TEST;": ["SELECTALL", "TABLE TEST", "SELECT 0 2", "STMT"],
    "SELECT * FROM A, B;": ["SELECTALL", "TABLE A", "TABLE B", "SELECT 0 3", "STMT"],
    "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b;": ["SELECTALL"] + NORTH_AB + ["SELECT 0 2", "STMT"],
    "SELECT f1, f2 FROM A INNER JOIN B ON A.id_a = B.id_b;": ["NAME f1", "NAME f2"] + NORTH_AB + ["SELECT 0 3", "STMT"],
    "SELECT f1, f2 FROM A INNER JOIN B ON A.id_a = B.id_b WHERE f1 = 123;":
        ["NAME f1", "NAME f2"] + NORTH_AB + ["NAME f1", "NUMBER 123", "CMP 4", "WHERE", "SELECT 0 4", "STMT"],
    "SELECT f1, f2 FROM A INNER JOIN B ON A.id_a = B.id_b WHERE 123 >= f1 AND f1 < 200;":
        ["NAME f1", "NAME f2"] + NORTH_AB + ["NUMBER 123", "NAME f1", "CMP 6", "NAME f1", "NUMBER 200", "CMP 1", "AND", "WHERE", "SELECT 0 4", "STMT"],
    "SELECT id FROM A WHERE f1 IS NULL;": ["NAME id", "TABLE A", "NAME f1", "ISNULL", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id, COUNT(*) FROM A GROUP BY id;": ["NAME id", "COUNTALL", "TABLE A", "NAME id", "GROUPBYLIST 1", "SELECT 0 4", "STMT"],
    "SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a;":
        ["NAME id_a", "COUNTALL"] + NORTH_AB + ["NAME id_a", "GROUPBYLIST 1", "SELECT 0 4", "STMT"],
    "SELECT COUNT(*) FROM A WHERE id > 1;": ["COUNTALL", "TABLE A", "NAME id", "NUMBER 1", "CMP 2", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT f1 FROM A WHERE f1 IN (124);": ["NAME f1", "TABLE A", "NAME f1", "NUMBER 124", "ISIN 1", "WHERE", "SELECT 0 3", "STMT"],
    # probes
    "SELECT COUNT(*) FROM A WHERE id_a > 100;": ["COUNTALL", "TABLE A", "NAME id_a", "NUMBER 100", "CMP 2", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id_a FROM A WHERE id_a > 100;": ["NAME id_a", "TABLE A", "NAME id_a", "NUMBER 100", "CMP 2", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT COUNT(*) FROM A;": ["COUNTALL", "TABLE A", "SELECT 0 2", "STMT"],
    "SELECT f1, COUNT(*) FROM A GROUP BY f1;": ["NAME f1", "COUNTALL", "TABLE A", "NAME f1", "GROUPBYLIST 1", "SELECT 0 4", "STMT"],
    "SELECT COUNT(*), id_a FROM A GROUP BY id_a;": ["COUNTALL", "NAME id_a", "TABLE A", "NAME id_a", "GROUPBYLIST 1", "SELECT 0 4", "STMT"],
    "SELECT f2 FROM B WHERE f2 > 1.0;": ["NAME f2", "TABLE B", "NAME f2", "FLOAT 1", "CMP 2", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b WHERE f2 < 2.0;":
        ["SELECTALL"] + NORTH_AB + ["NAME f2", "FLOAT 2", "CMP 1", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id_b, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_b;":
        ["NAME id_b", "COUNTALL"] + NORTH_AB + ["NAME id_b", "GROUPBYLIST 1", "SELECT 0 4", "STMT"],
    "SELECT f1 FROM A WHERE id_a NOT IN (3, 4);": ["NAME f1", "TABLE A", "NAME id_a", "NUMBER 3", "NUMBER 4", "ISNOTIN 2", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id_a FROM A WHERE f1 = NULL;": ["NAME id_a", "TABLE A", "NAME f1", "NULL", "CMP 4", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id_a FROM A WHERE 1 = 1;": ["NAME id_a", "TABLE A", "NUMBER 1", "NUMBER 1", "CMP 4", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id_a FROM A WHERE id_a = 3 OR f1 = 10;":
        ["NAME id_a", "TABLE A", "NAME id_a", "NUMBER 3", "CMP 4", "NAME f1", "NUMBER 10", "CMP 4", "OR", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id_a FROM A WHERE id_a = 3 XOR f1 = 30;":
        ["NAME id_a", "TABLE A", "NAME id_a", "NUMBER 3", "CMP 4", "NAME f1", "NUMBER 30", "CMP 4", "XOR", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT A.id_a FROM A WHERE A.id_a <> 3;": ["FIELDNAME A.id_a", "TABLE A", "FIELDNAME A.id_a", "NUMBER 3", "CMP 3", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT x.id_a FROM A AS x WHERE x.f1 >= 30;":
        ["FIELDNAME x.id_a", "TABLE A", "ALIAS x", "FIELDNAME x.f1", "NUMBER 30", "CMP 6", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id_a, f1 FROM A INNER JOIN B ON A.id_a = B.id_b AND f1 > 10;":
        ["NAME id_a", "NAME f1", "TABLE A", "TABLE B", "FIELDNAME A.id_a", "FIELDNAME B.id_b", "CMP 4", "NAME f1", "NUMBER 10", "CMP 2", "AND", "ONEXPR",
         "JOIN 1", "SELECT 0 3", "STMT"],
    "SELECT id_a FROM A WHERE f1 IS NOT NULL;": ["NAME id_a", "TABLE A", "NAME f1", "ISNOTNULL", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT f1 FROM A WHERE 10 < f1;": ["NAME f1", "TABLE A", "NUMBER 10", "NAME f1", "CMP 1", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id_a, id_b FROM A, B WHERE id_a < id_b;":
        ["NAME id_a", "NAME id_b", "TABLE A", "TABLE B", "NAME id_a", "NAME id_b", "CMP 1", "WHERE", "SELECT 0 5", "STMT"],
    "SELECT id_a, id_b FROM A INNER JOIN B ON B.id_b = A.id_a;":
        ["NAME id_a", "NAME id_b", "TABLE A", "TABLE B", "FIELDNAME B.id_b", "FIELDNAME A.id_a", "CMP 4", "ONEXPR", "JOIN 1", "SELECT 0 3", "STMT"],
    # three-way join, column orders
    "SELECT * FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c;":
        ["SELECTALL"] + NORTH_AB + ["TABLE C", "FIELDNAME A.id_a", "FIELDNAME C.id_c", "CMP 4", "ONEXPR", "JOIN 1", "SELECT 0 2", "STMT"],
    "SELECT * FROM T1 INNER JOIN T2 ON T1.a = T2.f;":
        ["SELECTALL", "TABLE T1", "TABLE T2", "FIELDNAME T1.a", "FIELDNAME T2.f", "CMP 4", "ONEXPR", "JOIN 1", "SELECT 0 2", "STMT"],
    "SELECT * FROM orders INNER JOIN cust ON orders.customer = cust.cust_id;":
        ["SELECTALL", "TABLE orders", "TABLE cust", "FIELDNAME orders.customer", "FIELDNAME cust.cust_id", "CMP 4", "ONEXPR", "JOIN 1", "SELECT 0 2", "STMT"],
    "SELECT * FROM W;": ["SELECTALL", "TABLE W", "SELECT 0 2", "STMT"],
    # DOUBLE join keys
    "SELECT * FROM A INNER JOIN B ON A.x = B.y;": ["SELECTALL"] + XY + ["SELECT 0 2", "STMT"],
    "SELECT fa, fb FROM A INNER JOIN B ON B.y = A.x WHERE fa > 1;":
        ["NAME fa", "NAME fb", "TABLE A", "TABLE B", "FIELDNAME B.y", "FIELDNAME A.x", "CMP 4", "ONEXPR", "JOIN 1", "NAME fa", "NUMBER 1", "CMP 2", "WHERE",
         "SELECT 0 4", "STMT"],
    "SELECT fa, COUNT(*) FROM A INNER JOIN B ON A.x = B.y GROUP BY fa;": ["NAME fa", "COUNTALL"] + XY + ["NAME fa", "GROUPBYLIST 1", "SELECT 0 4", "STMT"],
    "SELECT COUNT(*) FROM A INNER JOIN B ON A.x = B.y;": ["COUNTALL"] + XY + ["SELECT 0 2", "STMT"],
    "SELECT fa, fb FROM A INNER JOIN B ON A.x = B.y AND fb > 10;":
        ["NAME fa", "NAME fb", "TABLE A", "TABLE B", "FIELDNAME A.x", "FIELDNAME B.y", "CMP 4", "NAME fb", "NUMBER 10", "CMP 2", "AND", "ONEXPR", "JOIN 1",
         "SELECT 0 3", "STMT"],
    # config 1
    "SELECT v FROM T WHERE v > 500000;": ["NAME v", "TABLE T", "NAME v", "NUMBER 500000", "CMP 2", "WHERE", "SELECT 0 3", "STMT"],
    # typed tables
    "SELECT id, born FROM P WHERE id > 1;": ["NAME id", "NAME born", "TABLE P", "NAME id", "NUMBER 1", "CMP 2", "WHERE", "SELECT 0 4", "STMT"],
    "SELECT id, seen, ok FROM P WHERE born IS NOT NULL;":
        ["NAME id", "NAME seen", "NAME ok", "TABLE P", "NAME born", "ISNOTNULL", "WHERE", "SELECT 0 5", "STMT"],
    "SELECT id FROM P WHERE ok = TRUE;": ["NAME id", "TABLE P", "NAME ok", "BOOL 1", "CMP 4", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id FROM P WHERE ok <> TRUE;": ["NAME id", "TABLE P", "NAME ok", "BOOL 1", "CMP 3", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id, w FROM P WHERE w >= 1.5 AND ok = FALSE;":
        ["NAME id", "NAME w", "TABLE P", "NAME w", "FLOAT 1.5", "CMP 6", "NAME ok", "BOOL 0", "CMP 4", "AND", "WHERE", "SELECT 0 4", "STMT"],
    "SELECT id, qid FROM P INNER JOIN Q ON P.born = Q.day;": ["NAME id", "NAME qid"] + PQ + ["SELECT 0 3", "STMT"],
    "SELECT born, COUNT(*) FROM P INNER JOIN Q ON P.born = Q.day GROUP BY born;":
        ["NAME born", "COUNTALL"] + PQ + ["NAME born", "GROUPBYLIST 1", "SELECT 0 4", "STMT"],
    "SELECT COUNT(*) FROM P WHERE seen IS NULL;": ["COUNTALL", "TABLE P", "NAME seen", "ISNULL", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id FROM P WHERE born = '2001-02-03';": ["NAME id", "TABLE P", "NAME born", "STRING '2001-02-03'", "CMP 4", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id FROM P WHERE id = 'abc';": ["NAME id", "TABLE P", "NAME id", "STRING 'abc'", "CMP 4", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id, born FROM P;": ["NAME id", "NAME born", "TABLE P", "SELECT 0 3", "STMT"],
    "SELECT id, born, seen, ok FROM P;": ["NAME id", "NAME born", "NAME seen", "NAME ok", "TABLE P", "SELECT 0 5", "STMT"],
    "SELECT id FROM P;": ["NAME id", "TABLE P", "SELECT 0 2", "STMT"],
    # ... VARCHAR columns
    "SELECT id, name FROM P;": ["NAME id", "NAME name", "TABLE P", "SELECT 0 3", "STMT"],
    "SELECT id FROM P WHERE name = 'bob';": ["NAME id", "TABLE P", "NAME name", "STRING 'bob'", "CMP 4", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id FROM P WHERE name <> 'bob';": ["NAME id", "TABLE P", "NAME name", "STRING 'bob'", "CMP 3", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id FROM P WHERE name = 'nobody';": ["NAME id", "TABLE P", "NAME name", "STRING 'nobody'", "CMP 4", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id FROM P WHERE name <> 'nobody';": ["NAME id", "TABLE P", "NAME name", "STRING 'nobody'", "CMP 3", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id, name FROM P WHERE name IS NULL;": ["NAME id", "NAME name", "TABLE P", "NAME name", "ISNULL", "WHERE", "SELECT 0 4", "STMT"],
    "SELECT id FROM P WHERE name > 'ann';": ["NAME id", "TABLE P", "NAME name", "STRING 'ann'", "CMP 2", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT id FROM P WHERE name = 5;": ["NAME id", "TABLE P", "NAME name", "NUMBER 5", "CMP 4", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT qid, note FROM Q WHERE note = 'zz' OR note = 'x';":
        ["NAME qid", "NAME note", "TABLE Q", "NAME note", "STRING 'zz'", "CMP 4", "NAME note", "STRING 'x'", "CMP 4", "OR", "WHERE", "SELECT 0 4", "STMT"],
    "SELECT qid, note FROM Q;": ["NAME qid", "NAME note", "TABLE Q", "SELECT 0 3", "STMT"],
    "SELECT cid, vid FROM C INNER JOIN V ON C.city = V.town;":
        ["NAME cid", "NAME vid", "TABLE C", "TABLE V", "FIELDNAME C.city", "FIELDNAME V.town", "CMP 4", "ONEXPR", "JOIN 1", "SELECT 0 3", "STMT"],
    "SELECT city, COUNT(*) FROM C GROUP BY city;": ["NAME city", "COUNTALL", "TABLE C", "NAME city", "GROUPBYLIST 1", "SELECT 0 4", "STMT"],
    "SELECT city, COUNT(*) FROM C INNER JOIN V ON C.city = V.town GROUP BY city;":
        ["NAME city", "COUNTALL", "TABLE C", "TABLE V", "FIELDNAME C.city", "FIELDNAME V.town", "CMP 4", "ONEXPR", "JOIN 1", "NAME city", "GROUPBYLIST 1",
         "SELECT 0 4", "STMT"],
    "SELECT cid FROM C WHERE city IN ('rome');": ["NAME cid", "TABLE C", "NAME city", "STRING 'rome'", "ISIN 1", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT cid FROM C WHERE city NOT IN ('rome');": ["NAME cid", "TABLE C", "NAME city", "STRING 'rome'", "ISNOTIN 1", "WHERE", "SELECT 0 3", "STMT"],
    "SELECT k, v FROM N;": ["NAME k", "NAME v", "TABLE N", "SELECT 0 3", "STMT"],
}

N = r"(-?\d+)"
# randomised fixtures: one hand-written queue per template, the literals of the case filled in
TEMPLATES = [
    (r"SELECT f1 FROM (TEST|[DU]\d);", lambda t: ["NAME f1", f"TABLE {t}", "SELECT 0 2", "STMT"]),
    (rf"SELECT f1, f2 FROM A INNER JOIN B ON A\.id_a = B\.id_b WHERE f1 > {N} AND f2 <= {N};",
     lambda k, m: ["NAME f1", "NAME f2"] + NORTH_AB + ["NAME f1", f"NUMBER {k}", "CMP 2", "NAME f2", f"NUMBER {m}", "CMP 5", "AND", "WHERE", "SELECT 0 4", "STMT"]),
    (r"SELECT id_a, COUNT\(\*\) FROM A GROUP BY id_a;", lambda: ["NAME id_a", "COUNTALL", "TABLE A", "NAME id_a", "GROUPBYLIST 1", "SELECT 0 4", "STMT"]),
    (rf"SELECT COUNT\(\*\) FROM A WHERE f1 < {N} OR id_a = {N};",
     lambda k, m: ["COUNTALL", "TABLE A", "NAME f1", f"NUMBER {k}", "CMP 1", "NAME id_a", f"NUMBER {m}", "CMP 4", "OR", "WHERE", "SELECT 0 3", "STMT"]),
    (r"SELECT id_a, f1 FROM A WHERE f1 IS NOT NULL;", lambda: ["NAME id_a", "NAME f1", "TABLE A", "NAME f1", "ISNOTNULL", "WHERE", "SELECT 0 4", "STMT"]),
    (rf"SELECT id_a FROM A WHERE f1 <> {N} XOR id_a < {N};",
     lambda k, m: ["NAME id_a", "TABLE A", "NAME f1", f"NUMBER {k}", "CMP 3", "NAME id_a", f"NUMBER {m}", "CMP 1", "XOR", "WHERE", "SELECT 0 3", "STMT"]),
    (rf"SELECT id_a, id_b, f2 FROM A INNER JOIN B ON A\.id_a = B\.id_b AND f2 > {N};",
     lambda k: ["NAME id_a", "NAME id_b", "NAME f2", "TABLE A", "TABLE B", "FIELDNAME A.id_a", "FIELDNAME B.id_b", "CMP 4", "NAME f2", f"NUMBER {k}", "CMP 2",
                "AND", "ONEXPR", "JOIN 1", "SELECT 0 4", "STMT"]),
    (r"SELECT f1, COUNT\(\*\) FROM A INNER JOIN B ON A\.id_a = B\.id_b GROUP BY f1;",
     lambda: ["NAME f1", "COUNTALL"] + NORTH_AB + ["NAME f1", "GROUPBYLIST 1", "SELECT 0 4", "STMT"]),
    (rf"SELECT COUNT\(\*\) FROM A INNER JOIN B ON A\.id_a = B\.id_b WHERE f1 >= {N};",
     lambda k: ["COUNTALL"] + NORTH_AB + ["NAME f1", f"NUMBER {k}", "CMP 6", "WHERE", "SELECT 0 3", "STMT"]),
]


def handwritten(sql):
    """-> the hand-written token list for a fixture SELECT, or None when the statement has none."""
    if sql in FIXED:
        return FIXED[sql]
    for pat, fn in TEMPLATES:
        m = re.fullmatch(pat, sql)
        if m:
            return fn(*[int(g) if re.fullmatch(r"-?\d+", g) else g for g in m.groups()])
    return None
