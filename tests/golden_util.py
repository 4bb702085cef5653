"""Helpers shared by the golden-fixture tests."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def all_cases(*files):
    out = []
    for fn in files:
        out.extend(load(fn))
    return out


def table_arrays(case, tname):
    """-> (list of 8-byte numpy columns, list of NULL arrays or None) for one fixture table."""
    t = case["tables"][tname]
    types = ddl_types(case)[tname]
    cols = []
    for c, ty in zip(t["cols"], types):
        cols.append(np.array(c, dtype=np.float64 if ty == "DOUBLE" else np.int64))
    nulls = None
    if t["nulls"] is not None:
        nulls = [None if x is None else np.array(x, dtype=np.uint8) for x in t["nulls"]]
    return cols, nulls


def ddl_types(case):
    """{table: [INT|DOUBLE per column]} parsed from the fixture's CREATE statements."""
    out = {}
    for s in case["ddl"]:
        name = s.split()[2]
        body = s[s.index("(") + 1:s.rindex(")")]
        out[name] = [p.split()[1].upper().replace("INTEGER", "INT") for p in body.split(",")]
    return out


def ddl_columns(case):
    out = {}
    for s in case["ddl"]:
        name = s.split()[2]
        body = s[s.index("(") + 1:s.rindex(")")]
        out[name] = [p.split()[0] for p in body.split(",")]
    return out


# ---- DML scripts (tests/golden/dml.json) ---------------------------------------------------------------

def script_schema(sql):
    """CREATE TABLE T (a INT, x DOUBLE) -> ("T", ["a", "x"], ["INT", "DOUBLE"])."""
    name = sql.split()[2]
    body = sql[sql.index("(") + 1:sql.rindex(")")]
    cols = [p.split()[0] for p in body.split(",")]
    types = [p.split()[1].upper().replace("INTEGER", "INT") for p in body.split(",")]
    return name, cols, types


def insert_values(sql, types):
    """INSERT INTO T VALUES (1, NULL, 0.5); -> [1, None, 0.5] (python ints / floats / None)."""
    body = sql[sql.index("(", sql.upper().index("VALUES")) + 1:sql.rindex(")")]
    out = []
    for tok, ty in zip(body.split(","), types):
        tok = tok.strip()
        out.append(None if tok.upper() == "NULL" else (float(tok) if ty == "DOUBLE" else int(tok)))
    return out


def raw_cell(v):
    """python value -> the 8 raw bytes as int64 (what the fixtures store), None stays None."""
    if v is None:
        return None
    if isinstance(v, float):
        return int(np.array([v], dtype=np.float64).view(np.int64)[0])
    return int(v)
