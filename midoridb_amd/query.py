"""Python mirror of the reference's query API (include/mdb_query.h) over ctypes.

Names and call sequence follow the reference's own tests (reference
tests/engine/executor_select.c:47-66): database_open -> query_execute -> while
query_cur_step(...) == MIDORIDB_ROW: query_column_int64(...) -> query_free -> database_close.
"""
import ctypes
import time
from ctypes import POINTER, Structure, c_char, c_char_p, c_double, c_int, c_int64, c_size_t, c_uint64, c_void_p

import numpy as np

from .lib import load_library

MIDORIDB_OK, MIDORIDB_ERROR, MIDORIDB_INTERNAL, MIDORIDB_NOMEM, MIDORIDB_ROW = 0, 1, 2, 3, 4
ST_OK_WITH_RESULTS, ST_OK_EXECUTED, ST_ERROR = 0, 1, 2


class Database(Structure):
    # struct database { void *tables; pthread_mutex_t mutex; }  (pthread_mutex_t = 40 bytes on x86-64 glibc)
    _fields_ = [("tables", c_void_p), ("mutex", c_char * 40)]


class ResultSet(Structure):
    _fields_ = [("table", c_void_p), ("cursor_blk", c_void_p), ("cursor_offset", c_size_t)]


class QueryOutputError(Structure):
    _fields_ = [("message", c_char * 1024)]


CT_VARCHAR = 0        # reference enum COLUMN_TYPE (include/primitive/column.h:17-25)


class QueryOutput(Structure):
    _fields_ = [("status", c_int), ("results", ResultSet), ("error", QueryOutputError), ("n_rows_aff", c_size_t)]


QUERY_SYMBOLS = [
    "database_open", "database_close", "query_execute", "query_cur_step", "query_column_int64", "query_free",
    "mdb_query_execute_rpn", "query_column_double", "query_column_is_null", "query_column_count", "query_column_name",
    "query_column_type", "query_row_count", "query_column_data", "query_exec_ms", "query_joined_rows",
    "mdb_table_append_columns", "mdb_table_generate", "mdb_sql_to_rpn", "query_column_text", "mdb_result_text_at",
    "mdb_database_device", "mdb_database_set_dist", "mdb_database_results_on_device", "mdb_database_groups_any_order", "mdb_database_joins_eliminated", "query_column_data_device", "query_column_nulls_device", "mdb_table_generate_shard",
]


def _bind(lib):
    if getattr(lib, "_mdb_query_bound", False):
        return
    PDB, PRS, PQO = POINTER(Database), POINTER(ResultSet), POINTER(QueryOutput)
    lib.database_open.argtypes = [PDB]
    lib.database_open.restype = c_int
    lib.database_close.argtypes = [PDB]
    lib.database_close.restype = None
    lib.query_execute.argtypes = [PDB, c_char_p]
    lib.query_execute.restype = PQO
    lib.mdb_query_execute_rpn.argtypes = [PDB, c_char_p]
    lib.mdb_query_execute_rpn.restype = PQO
    lib.query_cur_step.argtypes = [PRS]
    lib.query_cur_step.restype = c_int
    lib.query_column_int64.argtypes = [PRS, c_int]
    lib.query_column_int64.restype = c_int64
    lib.query_column_text.argtypes = [PRS, c_int]
    lib.query_column_text.restype = c_char_p
    lib.mdb_result_text_at.argtypes = [PRS, c_int, c_uint64]
    lib.mdb_result_text_at.restype = c_char_p
    lib.query_column_double.argtypes = [PRS, c_int]
    lib.query_column_double.restype = c_double
    lib.query_column_is_null.argtypes = [PRS, c_int]
    lib.query_column_is_null.restype = ctypes.c_bool
    lib.query_column_count.argtypes = [PRS]
    lib.query_column_count.restype = c_int
    lib.query_column_name.argtypes = [PRS, c_int]
    lib.query_column_name.restype = c_char_p
    lib.query_column_type.argtypes = [PRS, c_int]
    lib.query_column_type.restype = c_int
    lib.query_row_count.argtypes = [PRS]
    lib.query_row_count.restype = c_uint64
    lib.query_column_data.argtypes = [PRS, c_int]
    lib.query_column_data.restype = POINTER(c_int64)
    lib.query_exec_ms.argtypes = [PRS]
    lib.query_exec_ms.restype = c_double
    lib.query_joined_rows.argtypes = [PRS]
    lib.query_joined_rows.restype = c_uint64
    lib.query_free.argtypes = [PQO]
    lib.query_free.restype = None
    lib.mdb_table_append_columns.argtypes = [PDB, c_char_p, c_int, c_uint64, POINTER(c_void_p), POINTER(c_void_p)]
    lib.mdb_table_append_columns.restype = c_int
    lib.mdb_table_generate.argtypes = [PDB, c_char_p, c_uint64, c_uint64, POINTER(c_uint64)]
    lib.mdb_table_generate.restype = c_int
    lib.mdb_database_device.argtypes = [PDB]
    lib.mdb_database_device.restype = c_void_p
    lib.mdb_database_set_dist.argtypes = [PDB, c_void_p]
    lib.mdb_database_set_dist.restype = c_int
    lib.mdb_database_results_on_device.argtypes = [PDB, c_int]
    lib.mdb_database_results_on_device.restype = c_int
    lib.mdb_database_groups_any_order.argtypes = [PDB, c_int]
    lib.mdb_database_groups_any_order.restype = c_int
    lib.mdb_database_joins_eliminated.argtypes = [PDB]
    lib.mdb_database_joins_eliminated.restype = ctypes.c_ulonglong
    lib.query_column_data_device.argtypes = [PRS, c_int]
    lib.query_column_data_device.restype = c_void_p
    lib.query_column_nulls_device.argtypes = [PRS, c_int]
    lib.query_column_nulls_device.restype = c_void_p
    lib.mdb_table_generate_shard.argtypes = [PDB, c_char_p, c_uint64, c_uint64, c_uint64, c_uint64, POINTER(c_uint64)]
    lib.mdb_table_generate_shard.restype = c_int
    lib._mdb_query_bound = True


class QueryError(RuntimeError):
    pass


class Result:
    """A finished SELECT: column names (reference order), rows as tuples of int64 (NULL cells read 0,
    exactly what query_column_int64() returns), plus NULL flags and timings."""

    def __init__(self, names, types, columns, nulls, exec_ms, joined_rows):
        self.names = names
        self.types = types
        self.columns = columns
        self.nulls = nulls
        self.exec_ms = exec_ms
        self.joined_rows = joined_rows

    @property
    def nrows(self):
        return len(self.columns[0]) if self.columns else 0

    def rows(self):
        """VARCHAR cells come as str (None for NULL), everything else as int (the 8-byte cell)"""
        return [tuple(c[i] if (c[i] is None or isinstance(c[i], str)) else int(c[i]) for c in self.columns) for i in range(self.nrows)]


class DB:
    def __init__(self):
        self.lib = load_library()
        _bind(self.lib)
        self.db = Database()
        rc = self.lib.database_open(ctypes.byref(self.db))
        if rc != MIDORIDB_OK:
            raise QueryError(f"database_open failed: {rc}")
        self._open = True

    def close(self):
        if self._open:
            self.lib.database_close(ctypes.byref(self.db))
            self._open = False

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- sharded mode handed in by the host ------------------------------------------------------
    def device_handle(self):
        """mdb_dev_ctx* of the database (created on first use)"""
        h = self.lib.mdb_database_device(ctypes.byref(self.db))
        if not h:
            raise QueryError("no usable HIP device")
        return c_void_p(h)

    def last_plan(self):
        """what the last join / GROUP BY operator of this database's device context did (mdb_dev_last_plan) -> dict"""
        from .dev import last_plan_of
        return last_plan_of(self.lib, self.device_handle())

    def counters(self):
        """running totals of this database's device context (mdb_dev_counters) -> dict"""
        from .dev import counters_of
        return counters_of(self.lib, self.device_handle())

    def set_dist(self, dist_handle):
        """the database takes ownership of an mdb_dist* built for device_handle()"""
        if self.lib.mdb_database_set_dist(ctypes.byref(self.db), dist_handle) != 0:
            raise QueryError("mdb_database_set_dist failed")

    def joins_eliminated(self):
        """tables of this database's SELECT statements so far that the catalog's statistics made unnecessary to join (mdb_database_joins_eliminated)"""
        return int(self.lib.mdb_database_joins_eliminated(ctypes.byref(self.db)))

    def results_on_device(self, on=True):
        """SELECT results stay in HBM until a consumer reads them (mdb_database_results_on_device)"""
        if self.lib.mdb_database_results_on_device(ctypes.byref(self.db), 1 if on else 0) != 0:
            raise QueryError("mdb_database_results_on_device failed")

    def groups_any_order(self, on=True):
        """GROUP BY over a join may return its groups in any order (mdb_database_groups_any_order)"""
        if self.lib.mdb_database_groups_any_order(ctypes.byref(self.db), 1 if on else 0) != 0:
            raise QueryError("mdb_database_groups_any_order failed")

    def query_device(self, sql, copy=True):
        """SELECT with results kept on the device -> (names, types, [torch tensors or None per column], rows, joined rows, exec ms).
        copy=False: no tensor is made (timing runs: the statement ends when its columns exist in HBM)."""
        import torch
        t0 = time.perf_counter()
        raw = self.lib.query_execute(ctypes.byref(self.db), sql.encode())
        self.last_call_ms = (time.perf_counter() - t0) * 1e3
        out, status = self._run(raw)
        rs = ctypes.byref(out.contents.results)
        nc, nrows = self.lib.query_column_count(rs), int(self.lib.query_row_count(rs))
        names = [self.lib.query_column_name(rs, c).decode() for c in range(nc)]
        types = [self.lib.query_column_type(rs, c) for c in range(nc)]
        cols = []
        for c in range(nc):
            p = self.lib.query_column_data_device(rs, c)
            if not copy or not p or not nrows:
                cols.append(None)
                continue
            dt = torch.float64 if types[c] == 3 else torch.int64

            class _View:
                __cuda_array_interface__ = {"shape": (nrows,), "typestr": "<f8" if types[c] == 3 else "<i8", "data": (int(p), False), "version": 2}
            cols.append(torch.as_tensor(_View(), device="cuda").clone().to(dt))
        self.last_device_nulls = []      # per column: the NULL flags as a bool tensor, or None when no cell is NULL (query_column_nulls_device)
        for c in range(nc):
            pn = self.lib.query_column_nulls_device(rs, c) if copy and nrows else None
            if not pn:
                self.last_device_nulls.append(None)
                continue
            words = (nrows + 63) // 64

            class _Bits:
                __cuda_array_interface__ = {"shape": (words,), "typestr": "<i8", "data": (int(pn), False), "version": 2}
            w = torch.as_tensor(_Bits(), device="cuda").clone()
            bits = (w.unsqueeze(1) >> torch.arange(64, device=w.device, dtype=torch.int64)) & 1
            self.last_device_nulls.append(bits.reshape(-1)[:nrows].bool())
        res = (names, types, cols, nrows, int(self.lib.query_joined_rows(rs)), float(self.lib.query_exec_ms(rs)))
        self.lib.query_free(out)
        return res

    # -- statements -------------------------------------------------------------------------
    def _run(self, out):
        if not out:
            raise QueryError("query_execute returned NULL")
        status = out.contents.status
        if status == ST_ERROR:
            msg = out.contents.error.message.decode(errors="replace")
            self.lib.query_free(out)
            raise QueryError(msg.strip())
        return out, status

    def execute(self, sql):
        """CREATE / INSERT -> rows affected."""
        out, status = self._run(self.lib.query_execute(ctypes.byref(self.db), sql.encode()))
        n = out.contents.n_rows_aff
        self.lib.query_free(out)
        return n

    def query(self, sql, rpn=False, step_cursor=False, with_table=None):
        """SELECT -> Result.  step_cursor=True walks the result with query_cur_step()/query_column_int64()
        exactly like the reference's tests; otherwise whole columns are copied at once.  with_table: a callable that is handed
        `results.table` (an address) while the result is alive - what a consumer that reads the reference's `struct table` itself sees
        (include/mdb_legacy.h); its return value is kept in Result.table_view."""
        fn = self.lib.mdb_query_execute_rpn if rpn else self.lib.query_execute
        t0 = time.perf_counter()
        raw = fn(ctypes.byref(self.db), sql.encode())
        self.last_call_ms = (time.perf_counter() - t0) * 1e3	# wall time of the C call alone (no Python-side copies)
        out, status = self._run(raw)
        if status != ST_OK_WITH_RESULTS:
            self.lib.query_free(out)
            raise QueryError("not a SELECT")
        rs = ctypes.byref(out.contents.results)
        nc = self.lib.query_column_count(rs)
        nrows = int(self.lib.query_row_count(rs))
        names = [self.lib.query_column_name(rs, c).decode() for c in range(nc)]
        types = [self.lib.query_column_type(rs, c) for c in range(nc)]
        cols, nulls = [], []
        if step_cursor:
            cols = [[] for _ in range(nc)]
            nulls = [[] for _ in range(nc)]
            while self.lib.query_cur_step(rs) == MIDORIDB_ROW:
                for c in range(nc):
                    if types[c] == CT_VARCHAR:
                        t = self.lib.query_column_text(rs, c)
                        cols[c].append(None if t is None else t.decode())
                    else:
                        cols[c].append(self.lib.query_column_int64(rs, c))
                    nulls[c].append(bool(self.lib.query_column_is_null(rs, c)))
            cols = [np.array(c, dtype=object if types[i] == CT_VARCHAR else np.int64) for i, c in enumerate(cols)]
            nulls = [np.array(x, dtype=bool) for x in nulls]
        else:
            for c in range(nc):
                if types[c] == CT_VARCHAR:       # dictionary ids -> strings
                    texts = [self.lib.mdb_result_text_at(rs, c, i) for i in range(nrows)]
                    cols.append(np.array([None if t is None else t.decode() for t in texts], dtype=object))
                    continue
                p = self.lib.query_column_data(rs, c)
                cols.append(np.ctypeslib.as_array(p, shape=(nrows,)).copy() if nrows else np.zeros(0, dtype=np.int64))
            nulls = [None] * nc
        res = Result(names, types, cols, nulls, float(self.lib.query_exec_ms(rs)), int(self.lib.query_joined_rows(rs)))
        if with_table is not None:
            res.table_view = with_table(out.contents.results.table)
        self.lib.query_free(out)
        return res

    # -- ingest -------------------------------------------------------------------------------
    def append_columns(self, table, cols, nulls=None):
        """cols: one sequence per table column - numbers, or str / None for a VARCHAR column"""
        n = len(cols[0])
        arrs, keep = [], []
        for c in cols:
            is_text = (c.dtype.kind in "OUS" and any(isinstance(v, str) for v in c)) if isinstance(c, np.ndarray) else \
                (len(c) and any(isinstance(v, str) for v in c))
            if is_text:                                           # VARCHAR: an array of char pointers (NULL = SQL NULL)
                bufs = [None if v is None else ctypes.create_string_buffer(v.encode()) for v in c]
                keep.append(bufs)
                arrs.append(np.array([0 if b is None else ctypes.addressof(b) for b in bufs], dtype=np.int64))
                continue
            a = np.asarray(c)
            arrs.append(np.ascontiguousarray(a.view(np.int64) if a.dtype == np.float64 else np.asarray(c, dtype=np.int64)))
        cp = (c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        narrs = None
        npp = None
        if nulls is not None and any(x is not None for x in nulls):
            narrs = [None if x is None else np.ascontiguousarray(x, dtype=np.uint8) for x in nulls]
            npp = (c_void_p * len(arrs))(*[(a.ctypes.data if a is not None else None) for a in narrs])
        rc = self.lib.mdb_table_append_columns(ctypes.byref(self.db), table.encode(), len(arrs), n, cp, npp)
        if rc != 0:
            raise QueryError(f"mdb_table_append_columns({table}) failed: {rc}")

    def generate_shard(self, table, n, first_index, domain, seed, modulus=None):
        mod = None
        if modulus is not None:
            mod = (c_uint64 * len(modulus))(*modulus)
        rc = self.lib.mdb_table_generate_shard(ctypes.byref(self.db), table.encode(), n, first_index, domain, seed, mod)
        if rc != 0:
            raise QueryError(f"mdb_table_generate_shard({table}) failed: {rc}")

    def generate(self, table, n, seed, modulus=None):
        mod = None
        if modulus is not None:
            mod = (c_uint64 * len(modulus))(*modulus)
        rc = self.lib.mdb_table_generate(ctypes.byref(self.db), table.encode(), n, seed, mod)
        if rc != 0:
            raise QueryError(f"mdb_table_generate({table}) failed: {rc}")
