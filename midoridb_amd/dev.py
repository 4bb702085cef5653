"""Python binding of the device C-ABI (include/mdb_dev.h) for torch tensors.

PyTorch is plumbing only: it owns device memory (tensor.data_ptr()) and the HIP stream the
library launches on, and torch.distributed provides the RCCL all-to-all in bench.py.  Every
operator below is one call through the C-ABI into hand-written HIP; nothing here computes.
"""
import ctypes
from ctypes import POINTER, byref, c_char_p, c_double, c_int, c_int32, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

import numpy as np
import torch

from .lib import load_library

MDB_ORDER_FIRST = 1
MDB_KEYS_MAY_ALIAS = 2
MDB_COUNTS_OPTIONAL = 4

# predicate opcodes / comparison codes / value types (include/mdb_dev.h)
P_CMP_COL_CONST, P_CMP_CONST_COL, P_CMP_COL_COL, P_ISNULL, P_CONST, P_AND, P_OR, P_XOR = 1, 2, 3, 4, 5, 6, 7, 8
CMP_LT, CMP_GT, CMP_NE, CMP_EQ, CMP_LE, CMP_GE = 1, 2, 3, 4, 5, 6
T_INT64, T_DOUBLE = 0, 1


class PredInsn(ctypes.Structure):
    _fields_ = [("op", c_int32), ("cmp", c_int32), ("type", c_int32), ("a", c_int32), ("b", c_int32),
                ("pad", c_int32), ("imm", c_int64)]


class ColBinding(ctypes.Structure):
    _fields_ = [("values", c_void_p), ("nullbits", c_void_p), ("rid", c_void_p)]


class PayloadRight(ctypes.Structure):     # struct mdb_dev_payload_right (include/mdb_dev.h)
    _fields_ = [("keys", c_void_p), ("nulls", c_void_p), ("rows", c_uint64), ("npay", c_int), ("pay_in", c_void_p * 2), ("out", c_void_p * 2)]


class SortKey(ctypes.Structure):
    _fields_ = [("values", c_void_p), ("nullbits", c_void_p), ("rid", c_void_p), ("type", c_int32), ("desc", c_int32)]


class GatherCol(ctypes.Structure):
    _fields_ = [("src", c_void_p), ("src_nullbits", c_void_p), ("rid", c_void_p), ("dst", c_void_p), ("dst_nullbits", c_void_p)]


class ProjectCol(ctypes.Structure):
    _fields_ = [("values", c_void_p), ("nullbits", c_void_p), ("out_values", POINTER(c_void_p)), ("out_nullbits", POINTER(c_void_p))]


class ProfEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("launches", c_uint32), ("total_ms", c_double)]


class ColStats(ctypes.Structure):
    """struct mdb_dev_col_stats: what a catalog knows about a key column (include/mdb_dev.h)"""
    _fields_ = [("min", c_int64), ("max", c_int64), ("rows", c_uint64), ("nulls", c_uint64), ("flags", c_uint64)]


COL_DISTINCT = 1     # MDB_COL_DISTINCT


class PlanInfo(ctypes.Structure):
    """struct mdb_dev_plan_info: what the last join / GROUP BY operator did"""
    _fields_ = [(k, ctypes.c_uint32) for k in ("key_form", "key_bits", "levels", "digits", "minmax_pruned", "semijoin", "any_order", "ranged_order",
                                                "multi_one_pass", "retries", "samples", "from_stats", "payload_form", "group_form", "arena_mib", "small_form", "keys_are_left_column", "counts_all_one", "groups_as_bits", "payload_tables")]


class ExplainRequest(ctypes.Structure):
    """struct mdb_dev_explain_request"""
    _fields_ = [("left", ColStats), ("right", ColStats), ("further_tables", c_uint32), ("further_rows", c_uint64 * 2), ("left_nulls_bitmap", c_uint32),
                ("as_sample", c_uint32), ("num_cus", c_uint32)]


def explain(op, left, right=None, further_rows=(), as_sample=False, left_nulls_bitmap=False, num_cus=256, lib=None, cells=1):
    """plans as data - what the operator WOULD run for key columns with these statistics, without a GPU (mdb_dev_explain_*).
    op: "join_group_count" | "group_count" | "join_payload" (cells payload columns); left / right: dicts with min, max, rows and
    optionally nulls, distinct -> plan dict"""
    lib = lib or load_library()
    _bind(lib)

    def st(d):
        return ColStats(int(d["min"]), int(d["max"]), int(d["rows"]), int(d.get("nulls", 0)), COL_DISTINCT if d.get("distinct") else 0)
    rq = ExplainRequest()
    rq.left = st(left)
    if right is not None:
        rq.right = st(right)
    rq.further_tables = len(further_rows)
    for i, r in enumerate(further_rows):
        rq.further_rows[i] = int(r)
    rq.left_nulls_bitmap = 1 if left_nulls_bitmap else 0
    rq.as_sample = 1 if as_sample else 0
    rq.num_cus = num_cus
    info = PlanInfo()
    if op == "join_payload":
        rc = lib.mdb_dev_explain_join_payload(byref(rq), int(cells), byref(info))
    else:
        fn = lib.mdb_dev_explain_join_group_count if op == "join_group_count" else lib.mdb_dev_explain_group_count
        rc = fn(byref(rq), byref(info))
    if rc != 0:
        raise RuntimeError(f"mdb_dev_explain_{op} failed ({rc})")
    return {k: int(getattr(info, k)) for k, _ in PlanInfo._fields_}


class Counters(ctypes.Structure):
    """struct mdb_dev_counters: running totals of a context"""
    _fields_ = [(k, ctypes.c_uint64) for k in ("operator_calls", "retries", "samples", "arena_grows", "alloc_misses")]


def counters_of(lib, handle):
    """mdb_dev_counters of a raw context handle -> dict"""
    _bind(lib)
    c = Counters()
    if lib.mdb_dev_counters(handle, byref(c)) != 0:
        raise RuntimeError("mdb_dev_counters failed")
    return {k: int(getattr(c, k)) for k, _ in Counters._fields_}


def last_plan_of(lib, handle):
    """mdb_dev_last_plan of a raw context handle (a DeviceCtx's, or a database's: DB.device_handle()) -> dict"""
    _bind(lib)
    info = PlanInfo()
    if lib.mdb_dev_last_plan(handle, byref(info)) != 0:
        raise RuntimeError("mdb_dev_last_plan failed")
    return {k: int(getattr(info, k)) for k, _ in PlanInfo._fields_}


def _bind(lib):
    if getattr(lib, "_mdb_dev_bound", False):
        return
    P = c_void_p
    sig = {
        "mdb_dev_ctx_create": ([c_int, P, POINTER(P)], c_int),
        "mdb_dev_ctx_destroy": ([P], None),
        "mdb_dev_ctx_set_stream": ([P, P], c_int),
        "mdb_dev_last_error": ([P], c_char_p),
        "mdb_dev_sync": ([P], c_int),
        "mdb_dev_device_count": ([], c_int),
        "mdb_dev_reserve": ([P, c_size_t], c_int),
        "mdb_dev_set_overlap": ([P, c_int], c_int),
        "mdb_dev_set_narrow_keys": ([P, c_int], c_int),
        "mdb_dev_last_join_narrow": ([P], c_int),
        "mdb_dev_last_join_filter": ([P], c_int),
        "mdb_dev_counters": ([P, POINTER(Counters)], c_int),
        "mdb_dev_explain_join_group_count": ([POINTER(ExplainRequest), POINTER(PlanInfo)], c_int),
        "mdb_dev_explain_group_count": ([POINTER(ExplainRequest), POINTER(PlanInfo)], c_int),
        "mdb_dev_explain_join_payload": ([POINTER(ExplainRequest), c_int, POINTER(PlanInfo)], c_int),
        "mdb_dev_distinct_scan": ([P, P, P, c_uint64, c_int64, c_uint64, P, POINTER(c_int)], c_int),
        "mdb_dev_call_stats": ([P, P, POINTER(ColStats), P, POINTER(ColStats)], c_int),
        "mdb_dev_last_plan": ([P, POINTER(PlanInfo)], c_int),
        "mdb_dev_last_pairs_identity": ([P], c_int),
        "mdb_dev_arena_bytes": ([P], c_size_t),
        "mdb_dev_alloc": ([P, c_size_t, POINTER(P)], c_int),
        "mdb_dev_free": ([P, P], c_int),
        "mdb_dev_alloc_size": ([P, P], ctypes.c_size_t),
        "mdb_dev_retain": ([P, P], c_int),
        "mdb_dev_holders": ([P, P], ctypes.c_uint),
        "mdb_dev_memset": ([P, P, c_int, c_size_t], c_int),
        "mdb_dev_host_alloc": ([c_size_t], P),
        "mdb_dev_host_free": ([P], None),
        "mdb_dev_h2d": ([P, P, P, c_size_t], c_int),
        "mdb_dev_d2h": ([P, P, P, c_size_t], c_int),
        "mdb_dev_prof_enable": ([P, c_int], c_int),
        "mdb_dev_prof_reset": ([P], c_int),
        "mdb_dev_prof_read": ([P, POINTER(ProfEntry), c_int, POINTER(c_int)], c_int),
        "mdb_dev_prof_symbols": ([P, c_char_p, c_char_p, c_size_t], c_int),
        "mdb_dev_filter": ([P, POINTER(PredInsn), c_int, POINTER(ColBinding), c_int, c_uint64, P, POINTER(c_uint64)], c_int),
        "mdb_dev_gather64": ([P, P, P, P, c_uint64, P, P], c_int),
        "mdb_dev_double_join_keys": ([P, P, P, P, c_uint64, P, P], c_int),
        "mdb_dev_gather_cols": ([P, POINTER(GatherCol), c_int, c_uint64], c_int),
        "mdb_dev_filter_project": ([P, POINTER(PredInsn), c_int, POINTER(ColBinding), c_int, c_uint64, POINTER(ProjectCol), c_int,
                                    POINTER(c_uint64)], c_int),
        "mdb_dev_gather32": ([P, P, P, c_uint64, P], c_int),
        "mdb_dev_iota32": ([P, P, c_uint64], c_int),
        "mdb_dev_scatter_set64": ([P, P, P, P, c_uint64, c_int64, c_int], c_int),
        "mdb_dev_sort_perm": ([P, POINTER(SortKey), c_int, c_uint64, P], c_int),
        "mdb_dev_topk_perm": ([P, POINTER(SortKey), c_int, c_uint64, c_uint64, P, POINTER(c_uint64)], c_int),
        "mdb_dev_distinct_sel": ([P, POINTER(SortKey), c_int, c_uint64, P, POINTER(c_uint64)], c_int),
        "mdb_dev_group_count_multi": ([P, POINTER(SortKey), c_int, c_uint64, P, P, c_uint64, POINTER(c_uint64)], c_int),
        "mdb_dev_join_pairs": ([P, P, P, c_uint64, P, P, c_uint64, POINTER(P), POINTER(P), POINTER(c_uint64)], c_int),
        "mdb_dev_join_keys": ([P, P, P, c_uint64, P, P, c_uint64, POINTER(P), POINTER(c_uint64)], c_int),
        "mdb_dev_join_keys_ordered": ([P, P, P, c_uint64, P, P, c_uint64, POINTER(P), POINTER(c_uint64), POINTER(c_int)], c_int),
        "mdb_dev_join_payload": ([P, P, P, c_uint64, P, P, c_uint64, POINTER(P), c_int, POINTER(P)], c_int),
        "mdb_dev_join_payload_multi": ([P, P, P, c_uint64, POINTER(PayloadRight), c_int, c_int64, c_int64], c_int),
        "mdb_dev_cross_pairs": ([P, c_uint64, c_uint64, P, P], c_int),
        "mdb_dev_group_count": ([P, P, P, c_uint64, c_uint32, P, P, c_uint64, POINTER(c_uint64)], c_int),
        "mdb_dev_group_count_keys": ([P, P, P, c_uint64, P, P, c_uint64, POINTER(c_uint64)], c_int),
        "mdb_dev_join_group_count": ([P, P, P, c_uint64, P, P, c_uint64, c_uint32, P, P, P, c_uint64,
                                      POINTER(c_uint64), POINTER(c_uint64)], c_int),
        "mdb_dev_combine_counts": ([P, P, P, P, P, c_uint64, P, P, POINTER(c_uint64)], c_int),
        "mdb_dev_join_group_count_begin": ([P, P, P, c_uint64, c_uint64], c_int),
        "mdb_dev_join_group_count_finish": ([P, P, P, c_uint64, c_uint32, P, P, P, c_uint64, POINTER(c_uint64), POINTER(c_uint64)], c_int),
        "mdb_dev_join_group_count_i32": ([P, P, c_uint64, P, c_uint64, c_uint32, P, P, P, c_uint64, POINTER(c_uint64), POINTER(c_uint64)], c_int),
        "mdb_dev_join_group_count_begin_i32": ([P, P, c_uint64, c_uint64], c_int),
        "mdb_dev_join_group_count_finish_i32": ([P, P, c_uint64, c_uint32, P, P, P, c_uint64, POINTER(c_uint64), POINTER(c_uint64)], c_int),
        "mdb_dev_partition_by_dest": ([P, P, P, c_uint64, c_uint32, c_int, P, P, POINTER(c_uint64)], c_int),
        "mdb_dev_partition_by_dest_pruned": ([P, P, P, c_uint64, c_uint32, c_int, c_int64, c_int64, c_int64, c_int64, P, P, POINTER(c_uint64)], c_int),
        "mdb_dev_key_range": ([P, P, P, c_uint64, POINTER(c_int64), POINTER(c_int64)], c_int),
        "mdb_dev_widen32to64": ([P, P, c_uint64, P], c_int),
        "mdb_dev_gen_keys": ([P, P, c_uint64, c_uint64, c_uint64, c_uint64, c_uint64], c_int),
        "mdb_dev_join_group_count_multi": ([P, P, P, c_uint64, c_int, POINTER(P), POINTER(P), POINTER(c_uint64), c_uint32, P, P, P, c_uint64,
                                            POINTER(c_uint64), POINTER(c_uint64)], c_int),
        "mdb_dev_gen_payload": ([P, P, c_uint64, c_uint64, c_uint64, c_int], c_int),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    lib._mdb_dev_bound = True


DEV_SYMBOLS = [
    "mdb_dev_ctx_create", "mdb_dev_ctx_destroy", "mdb_dev_ctx_set_stream", "mdb_dev_last_error", "mdb_dev_sync",
    "mdb_dev_device_count", "mdb_dev_reserve", "mdb_dev_set_overlap", "mdb_dev_set_narrow_keys", "mdb_dev_last_join_narrow", "mdb_dev_last_join_filter", "mdb_dev_call_stats", "mdb_dev_last_plan", "mdb_dev_reload_knobs", "mdb_dev_counters", "mdb_dev_distinct_scan", "mdb_dev_explain_join_group_count", "mdb_dev_explain_group_count", "mdb_dev_explain_join_payload", "mdb_dev_last_pairs_identity", "mdb_dev_arena_bytes", "mdb_dev_alloc", "mdb_dev_free", "mdb_dev_memset",
    "mdb_dev_host_alloc", "mdb_dev_host_free", "mdb_dev_h2d", "mdb_dev_d2h", "mdb_dev_prof_enable", "mdb_dev_prof_reset", "mdb_dev_prof_read", "mdb_dev_prof_symbols", "mdb_dev_filter",
    "mdb_dev_gather64", "mdb_dev_gather_cols", "mdb_dev_filter_project", "mdb_dev_double_join_keys", "mdb_dev_gather32", "mdb_dev_iota32", "mdb_dev_scatter_set64", "mdb_dev_sort_perm", "mdb_dev_topk_perm", "mdb_dev_distinct_sel", "mdb_dev_group_count_multi", "mdb_dev_join_pairs", "mdb_dev_join_keys", "mdb_dev_join_keys_ordered", "mdb_dev_join_payload", "mdb_dev_join_payload_multi", "mdb_dev_cross_pairs", "mdb_dev_alloc_size", "mdb_dev_retain", "mdb_dev_holders", "mdb_dev_map_ids",
    "mdb_dev_group_count", "mdb_dev_group_count_keys", "mdb_dev_join_group_count", "mdb_dev_join_group_count_multi", "mdb_dev_combine_counts", "mdb_dev_join_group_count_begin", "mdb_dev_join_group_count_finish",
    "mdb_dev_join_group_count_i32", "mdb_dev_join_group_count_begin_i32", "mdb_dev_join_group_count_finish_i32",
    "mdb_dev_partition_by_dest", "mdb_dev_partition_by_dest_pruned", "mdb_dev_key_range", "mdb_dev_widen32to64", "mdb_dev_gen_keys", "mdb_dev_gen_payload",
]


def pack_nullbits(nulls):
    """bool/uint8 array (1 = NULL) -> uint64 words, bit (i & 63) of word (i >> 6)."""
    nulls = np.asarray(nulls).astype(bool)
    n = nulls.shape[0]
    words = (n + 63) // 64
    padded = np.zeros(words * 64, dtype=np.uint8)
    padded[:n] = nulls
    return np.packbits(padded.reshape(words, 64), axis=1, bitorder="little").view(np.uint64).reshape(words).copy()


def unpack_nullbits(words, n):
    words = np.ascontiguousarray(words, dtype=np.uint64)
    bits = np.unpackbits(words.view(np.uint8), bitorder="little")
    return bits[:n].astype(bool)


def _ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


class DeviceError(RuntimeError):
    pass


class _LibBuffer:
    """__cuda_array_interface__ view of a device buffer owned by the library; torch keeps this object alive as long as a
    tensor made from it exists, and the buffer is handed back to the library's allocator afterwards."""

    def __init__(self, ctx, ptr, numel, dtype):
        self._ctx, self._ptr = ctx, ptr
        item = torch.empty(0, dtype=dtype).element_size()
        kind = "f" if dtype.is_floating_point else "i"
        self.__cuda_array_interface__ = {"shape": (int(numel),), "typestr": f"<{kind}{item}", "data": (int(ptr), False), "version": 2}

    def __del__(self):
        try:
            if self._ptr and self._ctx.h:
                self._ctx.lib.mdb_dev_free(self._ctx.h, c_void_p(self._ptr))
        except Exception:
            pass
        self._ptr = 0


class DeviceCtx:
    """One mdb_dev_ctx bound to a torch device and (by default) torch's current stream."""

    def __init__(self, device=0, use_torch_stream=True):
        if not torch.cuda.is_available():
            raise DeviceError("no HIP device visible: the MI355X path cannot run (there is no CPU fallback)")
        self.lib = load_library()
        _bind(self.lib)
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        stream = c_void_p(torch.cuda.current_stream(self.device).cuda_stream) if use_torch_stream else None
        h = c_void_p()
        rc = self.lib.mdb_dev_ctx_create(device, stream, byref(h))
        if rc != 0:
            raise DeviceError(f"mdb_dev_ctx_create failed with {rc}")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.mdb_dev_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            msg = self.lib.mdb_dev_last_error(self.h)
            raise DeviceError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    # ---- helpers ------------------------------------------------------------------------
    def to_dev(self, arr, dtype=None):
        a = np.ascontiguousarray(arr, dtype=dtype)
        if a.dtype == np.uint64:
            return torch.from_numpy(a.view(np.int64)).to(self.device)
        if a.dtype == np.uint32:
            return torch.from_numpy(a.view(np.int32)).to(self.device)
        return torch.from_numpy(a).to(self.device)

    def nullbits_dev(self, nulls):
        if nulls is None:
            return None
        return self.to_dev(pack_nullbits(nulls))

    def sync(self):
        self._chk(self.lib.mdb_dev_sync(self.h), "sync")

    def reserve(self, nbytes):
        self._chk(self.lib.mdb_dev_reserve(self.h, nbytes), "reserve")

    def set_overlap(self, on=True):
        self._chk(self.lib.mdb_dev_set_overlap(self.h, 1 if on else 0), "set_overlap")

    def last_plan(self):
        """what the last join / GROUP BY operator did (mdb_dev_last_plan) -> dict"""
        return last_plan_of(self.lib, self.h)

    def distinct(self, keys, nulls=None, key_range=None):
        """no non-NULL key twice? - measured with mdb_dev_distinct_scan over a fresh bitmap of the column's window (what the store does at ingest)"""
        lo, hi = key_range if key_range is not None else self.key_range(keys, nulls)
        if lo > hi:
            return True
        bits = hi - lo + 1
        seen = torch.zeros((bits + 31) // 32, dtype=torch.int32, device=self.device)
        tw = c_int(0)
        self._chk(self.lib.mdb_dev_distinct_scan(self.h, _ptr(keys), _ptr(nulls) if nulls is not None else None, keys.numel(), lo, bits, _ptr(seen), byref(tw)),
                  "distinct_scan")
        return tw.value == 0

    def counters(self):
        """running totals since the context was created (mdb_dev_counters) -> dict"""
        return counters_of(self.lib, self.h)

    def call_stats(self, keys_l=None, stats_l=None, keys_r=None, stats_r=None):
        """catalog statistics (min, max) of the key columns of the operator calls that follow - call_stats() with nothing ends it"""
        def st(col, v):      # v = (min, max) or (min, max, flags)
            return None if v is None else ColStats(int(v[0]), int(v[1]), col.numel(), 0, int(v[2]) if len(v) > 2 else 0)
        sl, sr = st(keys_l, stats_l), st(keys_r, stats_r)
        self._chk(self.lib.mdb_dev_call_stats(self.h, _ptr(keys_l) if sl is not None else None, byref(sl) if sl is not None else None,
                                              _ptr(keys_r) if sr is not None else None, byref(sr) if sr is not None else None), "call_stats")

    def last_join_narrow(self):
        return bool(self.lib.mdb_dev_last_join_narrow(self.h))

    def last_join_form(self):
        """0 wide (64-bit hashes), 1 narrow (32-bit hashes of a 2^32-wide window), 2 compact narrow (k-bit hashes of the
        sampled window, direct-address leaves)"""
        return int(self.lib.mdb_dev_last_join_narrow(self.h))

    def last_join_filter(self):
        """(bitmap, minmax): bitmap = 0 when the last join did not filter the left table through the right table's key bitmap,
        else 1 + log2(values per bit); minmax = whether the left table's first level pruned by the right table's key range"""
        v = int(self.lib.mdb_dev_last_join_filter(self.h))
        return v & 0xFF, bool(v & 0x100)

    def last_join_multi(self):
        """the last join_group_count_multi counted all its right tables in one pass (no chain of two-table operators)"""
        return bool(self.lib.mdb_dev_last_join_filter(self.h) & 0x400)

    def last_pairs_identity(self):
        """the last join_pairs matched every left row with exactly one right row: its left vector is 0, 1, 2 ..."""
        return bool(self.lib.mdb_dev_last_pairs_identity(self.h))

    def last_join_unordered(self):
        """the last join_group_count ran without row ids and ordering (flags without MDB_ORDER_FIRST, no first rows wanted)"""
        return bool(self.lib.mdb_dev_last_join_filter(self.h) & 0x800)

    def last_join_one_pass_4096(self):
        """the last ordered join_group_count took ONE 4096-digit pass per table (key windows of 2^24 ... 2^27 values: 4-byte row words
        for the left table, leaves of up to 2^15 values shared by two workgroups)"""
        return bool(self.lib.mdb_dev_last_join_filter(self.h) & 0x1000)

    def last_join_ranged_order(self):
        """the last one-level join wrote its group records straight into the ordering kernel's row-id ranges (no record list, no sort levels)"""
        return bool(self.lib.mdb_dev_last_join_filter(self.h) & 0x2000)

    def last_join_levels(self):
        """partition levels of the last join / GROUP BY operator's final attempt: 1 (wide direct-address leaves) or 2"""
        return 1 if int(self.lib.mdb_dev_last_join_filter(self.h)) & 0x1200 else 2

    def set_narrow_keys(self, mode):
        """32-bit hashes for int32-range join keys: 0 never, 1 sampled and verified (default), 2 always try."""
        self._chk(self.lib.mdb_dev_set_narrow_keys(self.h, int(mode)), "set_narrow_keys")

    def arena_bytes(self):
        return int(self.lib.mdb_dev_arena_bytes(self.h))

    # ---- profiling ------------------------------------------------------------------------
    def prof_enable(self, on=True):
        self._chk(self.lib.mdb_dev_prof_enable(self.h, 1 if on else 0), "prof_enable")

    def prof_reset(self):
        self._chk(self.lib.mdb_dev_prof_reset(self.h), "prof_reset")

    def prof_read(self):
        buf = (ProfEntry * 64)()
        n = c_int()
        self._chk(self.lib.mdb_dev_prof_read(self.h, buf, 64, byref(n)), "prof_read")
        return {buf[i].name.decode(): (int(buf[i].launches), float(buf[i].total_ms)) for i in range(n.value)}

    def prof_symbols(self, name):
        """kernel names as rocprofv3's kernel trace lists them (demangled by the library, without the argument list) launched under
        profiler name `name`"""
        buf = ctypes.create_string_buffer(16384)
        self._chk(self.lib.mdb_dev_prof_symbols(self.h, name.encode(), buf, len(buf)), "prof_symbols")
        return [x for x in buf.value.decode().split("\n") if x]

    # ---- operators ------------------------------------------------------------------------
    def gen_keys(self, n, first_index, domain, seed, modulus=0):
        out = torch.empty(n, dtype=torch.int64, device=self.device)
        self._chk(self.lib.mdb_dev_gen_keys(self.h, _ptr(out), n, first_index, domain, seed, modulus), "gen_keys")
        return out

    def gen_payload(self, n, first_index, seed, kind):
        """payload column of the synthetic tables: kind 0 INT64 (SplitMix64 >> 33), kind 1 DOUBLE in [0, 1)"""
        out = torch.empty(n, dtype=torch.float64 if kind == 1 else torch.int64, device=self.device)
        self._chk(self.lib.mdb_dev_gen_payload(self.h, _ptr(out), n, first_index, seed, kind), "gen_payload")
        return out

    def join_group_count(self, keys_l, null_l, keys_r, null_r, out=None, flags=MDB_ORDER_FIRST, want_first=True, no_copies=False, alias=False):
        """-> (keys[G], counts[G], first[G], joined_rows); tensors are views into `out` buffers.  want_first=False: the result
        query_execute() asks for - (key, COUNT) in first-occurrence order, no first-row column (first[G] is returned as None).
        alias: MDB_KEYS_MAY_ALIAS, as query_execute() calls the operator - when every left row is a group the keys returned ARE keys_l (not a
        copy); no_copies: MDB_COUNTS_OPTIONAL as well (mdb_dev_join_keys_ordered's call) - when every COUNT is 1 the counts come back as None
        (last_plan() says which)."""
        if no_copies:
            flags |= MDB_KEYS_MAY_ALIAS | MDB_COUNTS_OPTIONAL
        elif alias:
            flags |= MDB_KEYS_MAY_ALIAS
        n_l, n_r = keys_l.numel(), keys_r.numel()
        cap = max(n_l, 1)
        if out is None:
            out = (torch.empty(cap, dtype=torch.int64, device=self.device),
                   torch.empty(cap, dtype=torch.int64, device=self.device),
                   torch.empty(cap, dtype=torch.int32, device=self.device) if want_first else None)
        ok, oc, of = out[0], out[1], (out[2] if want_first else None)
        g, j = c_uint64(), c_uint64()
        self._chk(self.lib.mdb_dev_join_group_count(self.h, _ptr(keys_l), _ptr(null_l), n_l, _ptr(keys_r), _ptr(null_r), n_r,
                                                    flags, _ptr(ok), _ptr(oc), _ptr(of), cap, byref(g), byref(j)),
                  "join_group_count")
        G = g.value
        if no_copies or alias:
            p = self.last_plan()
            return (keys_l[:G] if p["keys_are_left_column"] else ok[:G]), (None if p["counts_all_one"] else oc[:G]), (of[:G] if of is not None else None), j.value
        return ok[:G], oc[:G], (of[:G] if of is not None else None), j.value

    def join_group_count_unordered(self, keys_l, null_l, keys_r, null_r, out=None):
        """the same operator without MDB_ORDER_FIRST and without first rows: groups in unspecified order -> (keys[G], counts[G], joined_rows)"""
        n_l, n_r = keys_l.numel(), keys_r.numel()
        cap = max(n_l, 1)
        if out is None:
            out = (torch.empty(cap, dtype=torch.int64, device=self.device), torch.empty(cap, dtype=torch.int64, device=self.device))
        ok, oc = out[0], out[1]
        g, j = c_uint64(), c_uint64()
        self._chk(self.lib.mdb_dev_join_group_count(self.h, _ptr(keys_l), _ptr(null_l), n_l, _ptr(keys_r), _ptr(null_r), n_r, 0, _ptr(ok), _ptr(oc),
                                                    None, cap, byref(g), byref(j)), "join_group_count (unordered)")
        return ok[:g.value], oc[:g.value], j.value

    def join_group_count_multi(self, keys_l, null_l, rights, out=None, flags=MDB_ORDER_FIRST):
        """rights: [(keys, nullbits or None), ...] (1 ... 3 tables joined to keys_l on the one key) -> (keys[G], counts[G], first[G], joined rows)"""
        n_l, nr = keys_l.numel(), len(rights)
        cap, (ok, oc, of) = self._gc_out(n_l, out)
        kr = (c_void_p * nr)(*[r[0].data_ptr() for r in rights])
        nb = (c_void_p * nr)(*[(r[1].data_ptr() if r[1] is not None else None) for r in rights])
        ns = (c_uint64 * nr)(*[r[0].numel() for r in rights])
        g, j = c_uint64(), c_uint64()
        self._chk(self.lib.mdb_dev_join_group_count_multi(self.h, _ptr(keys_l), _ptr(null_l), n_l, nr, kr, nb, ns, flags, _ptr(ok), _ptr(oc), _ptr(of),
                                                          cap, byref(g), byref(j)), "join_group_count_multi")
        return ok[:g.value], oc[:g.value], of[:g.value], j.value

    def join_group_count_multi_unordered(self, keys_l, null_l, rights, out=None):
        """the same without MDB_ORDER_FIRST and without first rows: groups in unspecified order -> (keys[G], counts[G], joined rows)"""
        n_l, nr = keys_l.numel(), len(rights)
        cap = max(n_l, 1)
        if out is None:
            out = (torch.empty(cap, dtype=torch.int64, device=self.device), torch.empty(cap, dtype=torch.int64, device=self.device))
        kr = (c_void_p * nr)(*[r[0].data_ptr() for r in rights])
        nb = (c_void_p * nr)(*[(r[1].data_ptr() if r[1] is not None else None) for r in rights])
        ns = (c_uint64 * nr)(*[r[0].numel() for r in rights])
        g, j = c_uint64(), c_uint64()
        self._chk(self.lib.mdb_dev_join_group_count_multi(self.h, _ptr(keys_l), _ptr(null_l), n_l, nr, kr, nb, ns, 0, _ptr(out[0]), _ptr(out[1]), None,
                                                          cap, byref(g), byref(j)), "join_group_count_multi (unordered)")
        return out[0][:g.value], out[1][:g.value], j.value

    def _gc_out(self, n_l, out):
        cap = max(n_l, 1)
        if out is None:
            out = (torch.empty(cap, dtype=torch.int64, device=self.device),
                   torch.empty(cap, dtype=torch.int64, device=self.device),
                   torch.empty(cap, dtype=torch.int32, device=self.device))
        return cap, out

    def join_group_count_begin(self, keys_l, null_l, n_r_max):
        """Split form: partition the left table now (no host sync) ...  int32 key tensors (the 4-byte wire format of
        the exchange) go to the _i32 entry points: no NULL bitmaps there."""
        if keys_l.dtype == torch.int32:
            if null_l is not None:
                raise ValueError("int32 key columns carry no NULL bitmap")
            self._chk(self.lib.mdb_dev_join_group_count_begin_i32(self.h, _ptr(keys_l), keys_l.numel(), n_r_max), "join_group_count_begin_i32")
        else:
            self._chk(self.lib.mdb_dev_join_group_count_begin(self.h, _ptr(keys_l), _ptr(null_l), keys_l.numel(), n_r_max),
                      "join_group_count_begin")
        self._pending_left = keys_l     # keep the tensor alive until finish()

    def join_group_count_finish(self, keys_r, null_r, out=None, flags=MDB_ORDER_FIRST):
        """... and complete with the right table; same results as join_group_count()."""
        n_l = self._pending_left.numel()
        cap, out = self._gc_out(n_l, out)
        ok, oc, of = out
        g, j = c_uint64(), c_uint64()
        if keys_r.dtype == torch.int32:
            if null_r is not None:
                raise ValueError("int32 key columns carry no NULL bitmap")
            self._chk(self.lib.mdb_dev_join_group_count_finish_i32(self.h, _ptr(keys_r), keys_r.numel(), flags, _ptr(ok), _ptr(oc), _ptr(of),
                                                                   min(cap, ok.numel()), byref(g), byref(j)), "join_group_count_finish_i32")
        else:
            self._chk(self.lib.mdb_dev_join_group_count_finish(self.h, _ptr(keys_r), _ptr(null_r), keys_r.numel(), flags, _ptr(ok),
                                                               _ptr(oc), _ptr(of), min(cap, ok.numel()), byref(g), byref(j)),
                      "join_group_count_finish")
        self._pending_left = None
        G = g.value
        return ok[:G], oc[:G], of[:G], j.value

    def join_group_count_i32(self, keys_l, keys_r, out=None, flags=MDB_ORDER_FIRST):
        """join_group_count() over int32 key columns (no NULLs); keys come back as int64."""
        cap, out = self._gc_out(keys_l.numel(), out)
        ok, oc, of = out
        g, j = c_uint64(), c_uint64()
        self._chk(self.lib.mdb_dev_join_group_count_i32(self.h, _ptr(keys_l), keys_l.numel(), _ptr(keys_r), keys_r.numel(), flags, _ptr(ok),
                                                        _ptr(oc), _ptr(of), min(cap, ok.numel()), byref(g), byref(j)), "join_group_count_i32")
        G = g.value
        return ok[:G], oc[:G], of[:G], j.value

    def group_count(self, keys, nulls, flags=MDB_ORDER_FIRST):
        n = keys.numel()
        cap = max(n, 1)
        of = torch.empty(cap, dtype=torch.int32, device=self.device)
        oc = torch.empty(cap, dtype=torch.int64, device=self.device)
        g = c_uint64()
        self._chk(self.lib.mdb_dev_group_count(self.h, _ptr(keys), _ptr(nulls), n, flags, _ptr(of), _ptr(oc), cap, byref(g)),
                  "group_count")
        return of[:g.value], oc[:g.value]

    def group_count_keys(self, keys, nulls, out=None):
        """GROUP BY + COUNT(*) as (keys[G], counts[G]) in unspecified order, or None when the form does not serve the column
        (mdb_dev_group_count_keys returns 1: the ordered operator answers)"""
        n = keys.numel()
        cap = max(n, 1)
        if out is None:
            out = (torch.empty(cap, dtype=torch.int64, device=self.device), torch.empty(cap, dtype=torch.int64, device=self.device))
        g = c_uint64()
        rc = self.lib.mdb_dev_group_count_keys(self.h, _ptr(keys), _ptr(nulls), n, _ptr(out[0]), _ptr(out[1]), cap, byref(g))
        if rc == 1:
            return None
        self._chk(rc, "group_count_keys")
        return out[0][:g.value], out[1][:g.value]

    def join_pairs(self, keys_l, null_l, keys_r, null_r):
        pl, pr, cnt = c_void_p(), c_void_p(), c_uint64()
        self._chk(self.lib.mdb_dev_join_pairs(self.h, _ptr(keys_l), _ptr(null_l), keys_l.numel(), _ptr(keys_r), _ptr(null_r),
                                              keys_r.numel(), byref(pl), byref(pr), byref(cnt)), "join_pairs")
        J = cnt.value
        if not J:
            self._chk(self.lib.mdb_dev_free(self.h, pl), "free")
            self._chk(self.lib.mdb_dev_free(self.h, pr), "free")
            return (torch.empty(0, dtype=torch.int32, device=self.device), torch.empty(0, dtype=torch.int32, device=self.device))
        # tensors over the library's own buffers (no copy); they go back to its allocator when the tensors are collected
        return self._adopt(pl, J, torch.int32), self._adopt(pr, J, torch.int32)

    def join_keys(self, keys_l, null_l, keys_r, null_r):
        """the join's key column alone, every key once per joined row, in unspecified order (mdb_dev_join_keys)"""
        pk, cnt = c_void_p(), c_uint64()
        self._chk(self.lib.mdb_dev_join_keys(self.h, _ptr(keys_l), _ptr(null_l), keys_l.numel(), _ptr(keys_r), _ptr(null_r), keys_r.numel(),
                                             byref(pk), byref(cnt)), "join_keys")
        if not cnt.value:
            if pk:
                self._chk(self.lib.mdb_dev_free(self.h, pk), "free")
            return torch.empty(0, dtype=torch.int64, device=self.device)
        return self._adopt(pk, cnt.value, torch.int64)

    def join_keys_ordered(self, keys_l, null_l, keys_r, null_r):
        """the join's key column in the reference's row order when it is a primary-key join (one row per key on either side), else None
        (mdb_dev_join_keys_ordered)"""
        pk, cnt, served = c_void_p(), c_uint64(), c_int()
        self._chk(self.lib.mdb_dev_join_keys_ordered(self.h, _ptr(keys_l), _ptr(null_l), keys_l.numel(), _ptr(keys_r), _ptr(null_r), keys_r.numel(),
                                                     byref(pk), byref(cnt), byref(served)), "join_keys_ordered")
        if not served.value:
            return None
        if served.value == 2:       # every left row joined exactly one right row: the joined rows' key column IS keys_l (nothing was copied)
            return keys_l[:cnt.value]
        if not cnt.value:
            if pk:
                self._chk(self.lib.mdb_dev_free(self.h, pk), "free")
            return torch.empty(0, dtype=torch.int64, device=self.device)
        return self._adopt(pk, cnt.value, torch.int64)

    def join_payload(self, keys_l, null_l, keys_r, null_r, payload):
        """payload: one or two tensors of the right table (8-byte cells) -> list of tensors of len(keys_l) cells, the partner's cell of
        every left row - or None when it is not an every-left-row-has-exactly-one-partner join (mdb_dev_join_payload)"""
        npay = len(payload)
        outs = [torch.empty(keys_l.numel(), dtype=p.dtype, device=self.device) for p in payload]
        pin = (c_void_p * npay)(*[p.data_ptr() for p in payload])
        pout = (c_void_p * npay)(*[o.data_ptr() for o in outs])
        rc = self.lib.mdb_dev_join_payload(self.h, _ptr(keys_l), _ptr(null_l), keys_l.numel(), _ptr(keys_r), _ptr(null_r), keys_r.numel(), pin, npay, pout)
        if rc == 1:
            return None
        self._chk(rc, "join_payload")
        return outs

    def join_payload_multi(self, keys_l, rights, key_min, key_max):
        """rights: list of (keys_r, [one or two payload tensors]) - several right tables joined with ONE left key column; [key_min, key_max]
        bounds every key of every table -> per table the list of carried columns (len(keys_l) cells each), or None when not served
        (mdb_dev_join_payload_multi)"""
        arr = (PayloadRight * len(rights))()
        outs = []
        for i, (kr, payload) in enumerate(rights):
            o = [torch.empty(keys_l.numel(), dtype=p.dtype, device=self.device) for p in payload]
            outs.append(o)
            arr[i].keys = kr.data_ptr()
            arr[i].nulls = None
            arr[i].rows = kr.numel()
            arr[i].npay = len(payload)
            for c, (pi, po) in enumerate(zip(payload, o)):
                arr[i].pay_in[c] = pi.data_ptr()
                arr[i].out[c] = po.data_ptr()
        rc = self.lib.mdb_dev_join_payload_multi(self.h, _ptr(keys_l), None, keys_l.numel(), arr, len(rights), int(key_min), int(key_max))
        if rc == 1:
            return None
        self._chk(rc, "join_payload_multi")
        return outs

    def cross_pairs(self, n_l, n_r):
        ol = torch.empty(n_l * n_r, dtype=torch.int32, device=self.device)
        orr = torch.empty(n_l * n_r, dtype=torch.int32, device=self.device)
        self._chk(self.lib.mdb_dev_cross_pairs(self.h, n_l, n_r, _ptr(ol), _ptr(orr)), "cross_pairs")
        return ol, orr

    def filter(self, prog, cols, n):
        """prog: list of (op, cmp, type, a, b, imm); cols: list of (values, nullbits, rid) tensors/None."""
        insns = (PredInsn * max(len(prog), 1))()
        for i, (op, cmp_, typ, a, b, imm) in enumerate(prog):
            if typ == T_DOUBLE and isinstance(imm, float):
                imm = int(np.float64(imm).view(np.int64))
            insns[i] = PredInsn(op, cmp_, typ, a, b, 0, int(imm))
        binds = (ColBinding * max(len(cols), 1))()
        for i, (v, nb, rid) in enumerate(cols):
            binds[i] = ColBinding(v.data_ptr(), nb.data_ptr() if nb is not None else None,
                                  rid.data_ptr() if rid is not None else None)
        sel = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        cnt = c_uint64()
        self._chk(self.lib.mdb_dev_filter(self.h, insns, len(prog), binds, len(cols), n, _ptr(sel), byref(cnt)), "filter")
        return sel[:cnt.value]

    def gather64(self, src, src_null, idx, n):
        dst = torch.empty(max(n, 1), dtype=src.dtype, device=self.device)
        dnull = None
        if src_null is not None:
            dnull = torch.zeros((n + 63) // 64 or 1, dtype=torch.int64, device=self.device)
        self._chk(self.lib.mdb_dev_gather64(self.h, _ptr(src), _ptr(src_null), _ptr(idx), n, _ptr(dst), _ptr(dnull)), "gather64")
        return dst[:n], dnull

    def gather_cols(self, cols, n):
        """Whole-result projection in one launch: cols = [(src, src_nullbits or None, rid or None), ...] ->
        [(values[n], nullbits words or None), ...]."""
        arr = (GatherCol * len(cols))()
        outs = []
        for i, (src, nb, rid) in enumerate(cols):
            dst = torch.empty(max(n, 1), dtype=src.dtype, device=self.device)
            dnull = torch.zeros((n + 63) // 64 or 1, dtype=torch.int64, device=self.device) if nb is not None else None
            arr[i] = GatherCol(src.data_ptr(), nb.data_ptr() if nb is not None else None, rid.data_ptr() if rid is not None else None,
                               dst.data_ptr(), dnull.data_ptr() if dnull is not None else None)
            outs.append((dst[:n], dnull))
        self._chk(self.lib.mdb_dev_gather_cols(self.h, arr, len(cols), n), "gather_cols")
        return outs

    def filter_project(self, prog, cols, n, proj):
        """Scan + WHERE + projection of one table: prog / cols as filter(); proj = [(values, nullbits or None), ...] ->
        (count, [(values[count], nullbits words or None), ...]): tensors over the library-allocated outputs themselves (no
        copy; the buffer goes back to the library's allocator when the tensor is released)."""
        insns = (PredInsn * max(len(prog), 1))()
        for i, (op, cmp_, typ, a, b, imm) in enumerate(prog):
            if typ == T_DOUBLE and isinstance(imm, float):
                imm = int(np.float64(imm).view(np.int64))
            insns[i] = PredInsn(op, cmp_, typ, a, b, 0, int(imm))
        binds = (ColBinding * max(len(cols), 1))()
        for i, (v, nb, rid) in enumerate(cols):
            binds[i] = ColBinding(v.data_ptr(), nb.data_ptr() if nb is not None else None, rid.data_ptr() if rid is not None else None)
        pcs = (ProjectCol * max(len(proj), 1))()
        ov = [c_void_p() for _ in proj]
        on = [c_void_p() for _ in proj]
        for i, (v, nb) in enumerate(proj):
            pcs[i] = ProjectCol(v.data_ptr(), nb.data_ptr() if nb is not None else None, ctypes.pointer(ov[i]), ctypes.pointer(on[i]))
        cnt = c_uint64()
        self._chk(self.lib.mdb_dev_filter_project(self.h, insns, len(prog), binds, len(cols), n, pcs, len(proj), byref(cnt)), "filter_project")
        m = cnt.value
        outs = []
        for i, (v, nb) in enumerate(proj):
            dst = self._adopt(ov[i], max(n, 1), v.dtype)
            dnull = self._adopt(on[i], (n + 63) // 64 or 1, torch.int64) if (nb is not None and on[i].value) else None
            outs.append((dst[:m], dnull[:(m + 63) // 64] if dnull is not None else None))
        return m, outs

    def _adopt(self, ptr, numel, dtype):
        """a tensor over a buffer the library allocated (mdb_dev_alloc): no copy; mdb_dev_free when the tensor goes away"""
        return torch.as_tensor(_LibBuffer(self, ptr.value, numel, dtype), device=self.device)

    def double_join_keys(self, src, src_null, idx=None):
        """DOUBLE join keys as words that compare like IEEE `==`: (int64 words, NULL bits with the NaN rows added)."""
        n = idx.numel() if idx is not None else src.numel()
        dst = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        dnull = torch.zeros((n + 63) // 64 or 1, dtype=torch.int64, device=self.device)
        self._chk(self.lib.mdb_dev_double_join_keys(self.h, _ptr(src), _ptr(src_null), _ptr(idx), n, _ptr(dst), _ptr(dnull)),
                  "double_join_keys")
        return dst[:n], dnull

    def gather32(self, src, idx):
        n = idx.numel()
        dst = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        self._chk(self.lib.mdb_dev_gather32(self.h, _ptr(src), _ptr(idx), n, _ptr(dst)), "gather32")
        return dst[:n]

    def scatter_set64(self, dst, dst_nulls, idx, value_bits, set_null=False):
        """UPDATE's device half: dst[idx[k]] = value (all rows when idx is None), NULL bits set or cleared."""
        n = idx.numel() if idx is not None else dst.numel()
        self._chk(self.lib.mdb_dev_scatter_set64(self.h, _ptr(dst), _ptr(dst_nulls), _ptr(idx), n, int(value_bits),
                                                 1 if set_null else 0), "scatter_set64")

    def _sort_keys(self, keys):
        arr = (SortKey * len(keys))()
        for i, (v, nb, rid, ty, desc) in enumerate(keys):
            arr[i].values = v.data_ptr()
            arr[i].nullbits = nb.data_ptr() if nb is not None else None
            arr[i].rid = rid.data_ptr() if rid is not None else None
            arr[i].type = ty
            arr[i].desc = 1 if desc else 0
        return arr

    def distinct_sel(self, keys, n):
        """DISTINCT: ascending stream positions of the first occurrence of every distinct key combination."""
        sel = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        m = c_uint64(0)
        self._chk(self.lib.mdb_dev_distinct_sel(self.h, self._sort_keys(keys), len(keys), n, _ptr(sel), byref(m)), "distinct_sel")
        return sel[:m.value]

    def combine_counts(self, cnt1, first1, idx, cnt2):
        """chained fused joins: (cnt1[idx] * cnt2, first1[idx] (or idx), sum)"""
        n = idx.numel()
        out = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        outf = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        tot = c_uint64(0)
        self._chk(self.lib.mdb_dev_combine_counts(self.h, _ptr(cnt1), _ptr(first1), _ptr(idx), _ptr(cnt2), n, _ptr(out), _ptr(outf), byref(tot)),
                  "combine_counts")
        return out[:n], outf[:n], tot.value

    def group_count_multi(self, keys, n):
        """GROUP BY several columns + COUNT(*): -> (first positions ascending, counts)."""
        first = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        count = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        g = c_uint64(0)
        self._chk(self.lib.mdb_dev_group_count_multi(self.h, self._sort_keys(keys), len(keys), n, _ptr(first), _ptr(count), n, byref(g)),
                  "group_count_multi")
        return first[:g.value], count[:g.value]

    def sort_perm(self, keys, n):
        """ORDER BY: keys = [(values, nullbits or None, rid or None, T_INT64 | T_DOUBLE, desc bool), ...] ->
        int32 tensor perm with perm[k] = stream position of the k-th row in sorted order (stable)."""
        arr = (SortKey * len(keys))()
        for i, (v, nb, rid, ty, desc) in enumerate(keys):
            arr[i].values = v.data_ptr()
            arr[i].nullbits = nb.data_ptr() if nb is not None else None
            arr[i].rid = rid.data_ptr() if rid is not None else None
            arr[i].type = ty
            arr[i].desc = 1 if desc else 0
        perm = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        self._chk(self.lib.mdb_dev_sort_perm(self.h, arr, len(keys), n, _ptr(perm)), "sort_perm")
        return perm[:n]

    def topk_perm(self, keys, n, k):
        """ORDER BY ... LIMIT k: the first k entries of sort_perm(keys, n) without sorting the table ->
        (int32 tensor of min(k, n) stream positions, rows that went through the sort)."""
        arr = (SortKey * len(keys))()
        for i, (v, nb, rid, ty, desc) in enumerate(keys):
            arr[i].values = v.data_ptr()
            arr[i].nullbits = nb.data_ptr() if nb is not None else None
            arr[i].rid = rid.data_ptr() if rid is not None else None
            arr[i].type = ty
            arr[i].desc = 1 if desc else 0
        k = min(k, n)
        perm = torch.empty(max(k, 1), dtype=torch.int32, device=self.device)
        cand = c_uint64(0)
        self._chk(self.lib.mdb_dev_topk_perm(self.h, arr, len(keys), n, k, _ptr(perm), ctypes.byref(cand)), "topk_perm")
        return perm[:k], cand.value

    def partition_by_dest(self, keys, nulls, n_dest, out=None, with_rid=False, keys32=False, keep=None, own=None):
        """-> (keys grouped by destination, counts per destination[, source row of every entry]).  keys32: the keys
        come out as int32 (4-byte wire format; the caller knows from key_range() that they fit).  keep = (lo, hi): rows
        whose key lies outside are dropped (the other table's global key range); own = (lo, hi): the range promised for this
        column - a key outside it is an error."""
        n = keys.numel()
        if out is None:
            out = torch.empty(max(n, 1), dtype=torch.int32 if keys32 else torch.int64, device=self.device)
        elif keys32 and out.dtype != torch.int32:
            out = out.view(torch.int32)
        rid = torch.empty(max(n, 1), dtype=torch.int32, device=self.device) if with_rid else None
        counts = (c_uint64 * n_dest)()
        i64 = (-(1 << 63), (1 << 63) - 1)
        keep, own = keep or i64, own or i64
        self._chk(self.lib.mdb_dev_partition_by_dest_pruned(self.h, _ptr(keys), _ptr(nulls), n, n_dest, 1 if keys32 else 0, int(keep[0]), int(keep[1]),
                                                            int(own[0]), int(own[1]), _ptr(out), _ptr(rid), counts), "partition_by_dest")
        counts = [int(c) for c in counts]
        if with_rid:
            return out[:sum(counts)], counts, rid[:sum(counts)]
        return out[:sum(counts)], counts

    def key_range(self, keys, nulls=None):
        """column statistics: (min, max) of the non-NULL keys"""
        lo, hi = c_int64(0), c_int64(0)
        self._chk(self.lib.mdb_dev_key_range(self.h, _ptr(keys), _ptr(nulls), keys.numel(), byref(lo), byref(hi)), "key_range")
        return lo.value, hi.value

    def widen32(self, src32, out=None):
        n = src32.numel()
        if out is None:
            out = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        self._chk(self.lib.mdb_dev_widen32to64(self.h, _ptr(src32), n, _ptr(out)), "widen32to64")
        return out[:n]
