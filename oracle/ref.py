"""ref.py - TEST INFRASTRUCTURE ONLY: Python access to oracle/_ref/libmidori_ref.so, the REAL
reference compiled from /root/reference by `make -C oracle ref` (see ref_harness.c).

SQL text is turned into the parser's RPN token queue by the product's own SQL front end
(libmidoridb_amd.so: mdb_sql_to_rpn) - the reference's flex/bison parser cannot be built here -
and everything behind the parser seam is the unmodified reference code.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libmidori_ref.so")
_LIB = None


def available():
    return os.path.exists(_PATH)


def _lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(_PATH, mode=ctypes.RTLD_LOCAL)
        P = ctypes.c_void_p
        L.ref_open.restype = P
        L.ref_close.argtypes = [P]
        L.ref_error.argtypes = [P]
        L.ref_error.restype = ctypes.c_char_p
        L.ref_exec_rpn.argtypes = [P, ctypes.c_char_p]
        L.ref_exec_rpn.restype = ctypes.c_int
        L.ref_bulk_insert.argtypes = [P, ctypes.c_char_p, ctypes.c_int, ctypes.c_int64, P, P]
        L.ref_bulk_insert.restype = ctypes.c_int
        L.ref_result_ncols.argtypes = [P]
        L.ref_result_ncols.restype = ctypes.c_int
        L.ref_result_colname.argtypes = [P, ctypes.c_int]
        L.ref_result_colname.restype = ctypes.c_char_p
        L.ref_result_coltype.argtypes = [P, ctypes.c_int]
        L.ref_result_coltype.restype = ctypes.c_int
        L.ref_result_fetch.argtypes = [P, P, P, ctypes.c_int64]
        L.ref_result_fetch.restype = ctypes.c_int64
        L.ref_table_rows.argtypes = [P, ctypes.c_char_p]
        L.ref_table_rows.restype = ctypes.c_int64
        L.ref_rows_affected.argtypes = [P]
        L.ref_rows_affected.restype = ctypes.c_int64
        L.ref_table_fetch.argtypes = [P, ctypes.c_char_p, P, P, ctypes.c_int64]
        L.ref_table_fetch.restype = ctypes.c_int64
        L.ref_table_ncols.argtypes = [P, ctypes.c_char_p]
        L.ref_table_ncols.restype = ctypes.c_int
        if hasattr(L, "ref_foreign_fetch"):
            L.ref_foreign_ncols.argtypes = [P]
            L.ref_foreign_ncols.restype = ctypes.c_int
            L.ref_foreign_colname.argtypes = [P, ctypes.c_int]
            L.ref_foreign_colname.restype = ctypes.c_char_p
            L.ref_foreign_coltype.argtypes = [P, ctypes.c_int]
            L.ref_foreign_coltype.restype = ctypes.c_int
            L.ref_foreign_is_count.argtypes = [P, ctypes.c_int]
            L.ref_foreign_is_count.restype = ctypes.c_int
            L.ref_foreign_name.argtypes = [P]
            L.ref_foreign_name.restype = ctypes.c_char_p
            L.ref_foreign_fetch.argtypes = [P, P, P, ctypes.c_int64]
            L.ref_foreign_fetch.restype = ctypes.c_int64
            L.ref_foreign_text.argtypes = [ctypes.c_int64]
            L.ref_foreign_text.restype = ctypes.c_char_p
            if hasattr(L, "ref_foreign_precision"):
                L.ref_foreign_precision.argtypes = [P, ctypes.c_int]
                L.ref_foreign_precision.restype = ctypes.c_int
                L.ref_foreign_text_cell_ok.argtypes = [ctypes.c_int64, ctypes.c_int]
                L.ref_foreign_text_cell_ok.restype = ctypes.c_int
            L.ref_foreign_cursor_walk.argtypes = [P, ctypes.c_int, P, ctypes.c_int64]
            L.ref_foreign_cursor_walk.restype = ctypes.c_int64
        _LIB = L
    return _LIB


def foreign_table(table_ptr, max_rows=1 << 16):
    """A `struct table *` that came from the PRODUCT (the legacy view in front of a result, include/mdb_legacy.h) read with the reference's
    compiled layouts -> (table name, [(column name, type, is_count)], rows as a list of tuples of (value, is_null); VARCHAR values as str)."""
    L = _lib()
    nc = L.ref_foreign_ncols(table_ptr)
    cols = [(L.ref_foreign_colname(table_ptr, i).decode(), L.ref_foreign_coltype(table_ptr, i), bool(L.ref_foreign_is_count(table_ptr, i))) for i in range(nc)]
    n = L.ref_foreign_fetch(table_ptr, None, None, 0)
    vals = np.zeros((max(n, 1), max(nc, 1)), dtype=np.int64)
    nulls = np.zeros((max(n, 1), max(nc, 1)), dtype=np.uint8)
    got = L.ref_foreign_fetch(table_ptr, vals.ctypes.data_as(ctypes.c_void_p), nulls.ctypes.data_as(ctypes.c_void_p), max(n, 1))
    assert got == n
    rows = []
    for r in range(n):
        row = []
        for c in range(nc):
            v = int(vals[r, c])
            if cols[c][1] == 0:     # CT_VARCHAR: the cell is a pointer to `precision` bytes (read whole, as upstream's row copy does)
                if hasattr(L, "ref_foreign_precision"):
                    assert L.ref_foreign_text_cell_ok(v, L.ref_foreign_precision(table_ptr, c)) == 1, (r, c)
                v = L.ref_foreign_text(v).decode()
            row.append((v, bool(nulls[r, c])))
        rows.append(tuple(row))
    return L.ref_foreign_name(table_ptr).decode(), cols, rows


def foreign_cursor_walk(table_ptr, col, cap=1 << 16):
    """column `col` of a foreign table through the reference's cursor arithmetic (query_cur_step + query_column_int64)"""
    L = _lib()
    out = np.zeros(cap, dtype=np.int64)
    n = L.ref_foreign_cursor_walk(table_ptr, col, out.ctypes.data_as(ctypes.c_void_p), cap)
    return out[: min(n, cap)].tolist(), n


def sql_to_rpn(sql):
    """SQL -> RPN lines through the product's front end (midoridb_amd/csrc/mdb_sql.c)."""
    from midoridb_amd.lib import load_library
    lib = load_library()
    lib.mdb_sql_to_rpn.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    lib.mdb_sql_to_rpn.restype = ctypes.c_int
    buf = ctypes.create_string_buffer(1 << 16)
    err = ctypes.create_string_buffer(1024)
    rc = lib.mdb_sql_to_rpn(sql.encode(), buf, len(buf), err, len(err))
    if rc != 0:
        raise ValueError(f"SQL parse error: {err.value.decode()}")
    return buf.value.decode()


class RefError(RuntimeError):
    pass


class RefDB:
    """One reference database (struct database) driven behind its parser seam."""

    def __init__(self):
        self.L = _lib()
        self.h = self.L.ref_open()
        if not self.h:
            raise RefError("ref_open failed")

    def close(self):
        if self.h:
            self.L.ref_close(self.h)
            self.h = None

    def exec_rpn(self, rpn):
        rc = self.L.ref_exec_rpn(self.h, rpn.encode())
        if rc < 0:
            raise RefError(self.L.ref_error(self.h).decode().strip())
        return rc

    def execute(self, sql):
        """DDL/DML -> 0, SELECT -> 1 (result then available through fetch())."""
        return self.exec_rpn(sql_to_rpn(sql))

    def create_int_table(self, name, cols):
        self.execute(f"CREATE TABLE {name} ({', '.join(c + ' INT' for c in cols)});")

    def bulk_insert(self, table, cols, nulls=None):
        """cols: list of 8-byte arrays (int64 or float64), column-major; nulls: list of bool arrays or None."""
        ncols = len(cols)
        n = len(cols[0]) if ncols else 0
        vals = np.ascontiguousarray(np.stack([np.asarray(c).view(np.int64) if np.asarray(c).dtype == np.float64
                                              else np.asarray(c, dtype=np.int64) for c in cols]))
        nl = None
        if nulls is not None and any(x is not None for x in nulls):
            nl = np.ascontiguousarray(np.stack([np.zeros(n, dtype=np.uint8) if x is None else np.asarray(x, dtype=np.uint8)
                                                for x in nulls]))
        rc = self.L.ref_bulk_insert(self.h, table.encode(), ncols, n, vals.ctypes.data_as(ctypes.c_void_p),
                                    nl.ctypes.data_as(ctypes.c_void_p) if nl is not None else None)
        if rc != 0:
            raise RefError(f"ref_bulk_insert({table}) failed: {rc}")

    def fetch(self):
        """-> (column names in physical order, values int64 [nrows, ncols], nulls bool [nrows, ncols])."""
        nc = self.L.ref_result_ncols(self.h)
        if nc < 0:
            raise RefError("no result")
        names = [self.L.ref_result_colname(self.h, i).decode() for i in range(nc)]
        n = self.L.ref_result_fetch(self.h, None, None, 0)
        vals = np.zeros((max(n, 1), max(nc, 1)), dtype=np.int64)
        nulls = np.zeros((max(n, 1), max(nc, 1)), dtype=np.uint8)
        if n:
            self.L.ref_result_fetch(self.h, vals.ctypes.data_as(ctypes.c_void_p), nulls.ctypes.data_as(ctypes.c_void_p), n)
        return names, vals[:n, :nc], nulls[:n, :nc].astype(bool)

    def query(self, sql):
        rc = self.execute(sql)
        if rc != 1:
            raise RefError("not a SELECT")
        names, vals, _nulls = self.fetch()
        # values only, as query_column_int64() would return them: the reference's NULL bitmap is not
        # reliable in results (see ref_harness.c), a NULL cell reads as 0.  A VARCHAR cell is a heap pointer in THIS
        # process (src/primitive/column.c:255-293): it is followed here, a NULL cell (pointer 0) reads as None.
        self.L.ref_result_coltype.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self.L.ref_result_coltype.restype = ctypes.c_int
        types = [self.L.ref_result_coltype(self.h, k) for k in range(len(names))]

        def cell(i, k):
            v = int(vals[i, k])
            if types[k] != 0:
                return v
            return None if v == 0 else ctypes.string_at(v).decode()
        return names, [tuple(cell(i, k) for k in range(len(names))) for i in range(len(vals))]

    def rows_affected(self):
        return int(self.L.ref_rows_affected(self.h))

    def table_dump(self, name):
        """Live rows of a base table in scan order -> (values int64 [n, ncols], nulls bool [n, ncols])."""
        nc = self.L.ref_table_ncols(self.h, name.encode())
        n = self.L.ref_table_fetch(self.h, name.encode(), None, None, 0)
        if nc < 0 or n < 0:
            raise RefError(f"no table {name}")
        vals = np.zeros((max(n, 1), nc), dtype=np.int64)
        nulls = np.zeros((max(n, 1), nc), dtype=np.uint8)
        if n:
            self.L.ref_table_fetch(self.h, name.encode(), vals.ctypes.data_as(ctypes.c_void_p),
                                   nulls.ctypes.data_as(ctypes.c_void_p), n)
        return vals[:n], nulls[:n].astype(bool)

    def table_rows(self, name):
        return int(self.L.ref_table_rows(self.h, name.encode()))
