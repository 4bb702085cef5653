"""Python binding of the multi-GPU C-ABI (include/mdb_dist.h): the exchange itself - partition by destination,
counts, uneven all-to-all over RCCL, local join - runs inside libmidoridb_amd.so; nothing here moves keys.

The communicator id (128 bytes, created by rank 0 in C) has to reach the other ranks: `DistCtx.from_torch()` ships
it through an already initialised torch.distributed process group, `DistCtx.from_file()` through a file, and
`DistCtx(dev, world, rank, id_bytes)` takes it from wherever the host program got it.

`DistCtx.with_transport()` plugs another fabric in through struct mdb_dist_transport (callbacks): the tests use it to
run two ranks on ONE GPU with gloo carrying the bytes through host memory.
"""
import ctypes
from ctypes import CFUNCTYPE, POINTER, Structure, byref, c_char_p, c_double, c_int, c_size_t, c_uint64, c_void_p

import torch

from .lib import load_library

MDB_DIST_ID_BYTES = 128
WIRE_AUTO, WIRE_64, WIRE_32 = 0, 1, 2

_COUNTS = CFUNCTYPE(c_int, c_void_p, POINTER(c_uint64), POINTER(c_uint64), c_int)
_A2AV = CFUNCTYPE(c_int, c_void_p, c_void_p, POINTER(c_size_t), POINTER(c_size_t), c_void_p, POINTER(c_size_t), POINTER(c_size_t),
                  c_size_t, c_void_p)
_ALLRED = CFUNCTYPE(c_int, c_void_p, POINTER(c_uint64), c_int)
_DESTROY = CFUNCTYPE(None, c_void_p)


class Transport(Structure):
    _fields_ = [("self", c_void_p), ("counts", _COUNTS), ("alltoallv", _A2AV), ("allreduce_sum_u64", _ALLRED), ("destroy", _DESTROY)]


class DistCol(Structure):
    """struct mdb_dist_col: 8-byte cells, optional NULL bits, optional row-id vector (device pointers)."""
    _fields_ = [("values", c_void_p), ("nullbits", c_void_p), ("rid", c_void_p)]


class PlanInfo(Structure):
    """struct mdb_dist_plan_info: what the last regions-on-the-wire call planned"""
    _fields_ = [("world", ctypes.c_uint32), ("tables", ctypes.c_uint32), ("digit_bits", ctypes.c_uint32), ("digits_per_rank", ctypes.c_uint32),
                ("key_bits", ctypes.c_uint32), ("receiver_bits", ctypes.c_uint32), ("leaf_bits", ctypes.c_uint32), ("word_bytes", ctypes.c_uint32),
                ("completed", ctypes.c_uint32), ("region_words", ctypes.c_uint32 * 4), ("block_bytes", c_uint64 * 4), ("bytes_per_peer", c_uint64)]


KEEP_NULL_KEYS, NO_WAIT = 1, 2
DIST_PHASES = 4

DIST_SYMBOLS = [
    "mdb_dist_unique_id", "mdb_dist_id_via_file", "mdb_dist_init", "mdb_dist_destroy", "mdb_dist_world", "mdb_dist_rank",
    "mdb_dist_last_error", "mdb_dist_init_transport", "mdb_dist_set_wire", "mdb_dist_last_wire32", "mdb_dist_join_group_count",
    "mdb_dist_join_group_count_alloc", "mdb_dist_last_received_left", "mdb_dist_allreduce_sum_u64", "mdb_dist_barrier",
    "mdb_dist_set_key_ranges", "mdb_dist_last_pruned", "mdb_dist_last_fused", "mdb_dist_join_group_count_multi_alloc", "mdb_dist_group_count_keys_alloc", "mdb_dist_allgather_u64", "mdb_dist_shuffle_rows", "mdb_dist_wait_transfers", "mdb_dist_join_pairs",
    "mdb_dist_last_plan", "mdb_dist_set_phase_timing", "mdb_dist_last_phases", "mdb_dist_plan_preview", "mdb_dist_broadcast_rows", "mdb_dist_allgather_bytes",
]


def _bind(lib):
    if getattr(lib, "_mdb_dist_bound", False):
        return
    P = c_void_p
    sig = {
        "mdb_dist_unique_id": ([P], c_int),
        "mdb_dist_id_via_file": ([c_char_p, c_int, c_int, c_double, P], c_int),
        "mdb_dist_init": ([P, c_int, c_int, P, POINTER(P)], c_int),
        "mdb_dist_destroy": ([P], None),
        "mdb_dist_world": ([P], c_int),
        "mdb_dist_rank": ([P], c_int),
        "mdb_dist_last_error": ([P], c_char_p),
        "mdb_dist_init_transport": ([P, c_int, c_int, POINTER(Transport), POINTER(P)], c_int),
        "mdb_dist_set_wire": ([P, c_int], c_int),
        "mdb_dist_last_wire32": ([P], c_int),
        "mdb_dist_last_pruned": ([P], c_int),
        "mdb_dist_last_fused": ([P], c_int),
        "mdb_dist_last_plan": ([P, POINTER(PlanInfo)], c_int),
        "mdb_dist_plan_preview": ([c_int, c_int, POINTER(c_uint64), POINTER(ctypes.c_int64), POINTER(ctypes.c_int64), POINTER(PlanInfo)], c_int),
        "mdb_dist_set_phase_timing": ([P, c_int], c_int),
        "mdb_dist_last_phases": ([P, POINTER(c_double)], c_int),
        "mdb_dist_group_count_keys_alloc": ([P, P, P, c_uint64, POINTER(P), POINTER(P), POINTER(c_uint64)], c_int),
        "mdb_dist_join_group_count_multi_alloc": ([P, P, P, c_uint64, c_int, POINTER(P), POINTER(P), POINTER(c_uint64), POINTER(P), POINTER(P),
                                                   POINTER(c_uint64), POINTER(c_uint64)], c_int),
        "mdb_dist_set_key_ranges": ([P, POINTER(ctypes.c_int64), POINTER(ctypes.c_int64)], c_int),
        "mdb_dist_join_group_count": ([P, P, P, c_uint64, P, P, c_uint64, P, P, c_uint64, POINTER(c_uint64), POINTER(c_uint64)], c_int),
        "mdb_dist_join_group_count_alloc": ([P, P, P, c_uint64, P, P, c_uint64, ctypes.c_uint32, POINTER(P), POINTER(P), POINTER(P), POINTER(c_uint64),
                                             POINTER(c_uint64)], c_int),
        "mdb_dist_last_received_left": ([P], c_uint64),
        "mdb_dist_allreduce_sum_u64": ([P, POINTER(c_uint64), c_int], c_int),
        "mdb_dist_barrier": ([P], c_int),
        "mdb_dist_allgather_u64": ([P, POINTER(c_uint64), c_int, POINTER(c_uint64)], c_int),
        "mdb_dist_shuffle_rows": ([P, P, P, c_uint64, ctypes.c_uint32, POINTER(DistCol), c_int, POINTER(P), POINTER(P), POINTER(c_uint64)], c_int),
        "mdb_dist_wait_transfers": ([P], c_int),
        "mdb_dist_broadcast_rows": ([P, c_uint64, POINTER(DistCol), c_int, POINTER(P), POINTER(P), POINTER(c_uint64)], c_int),
        "mdb_dist_join_pairs": ([P, P, P, c_uint64, POINTER(DistCol), c_int, P, P, c_uint64, POINTER(DistCol), c_int, POINTER(P), POINTER(P),
                                 POINTER(P), POINTER(P), POINTER(P), POINTER(c_uint64)], c_int),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    lib._mdb_dist_bound = True


class DistError(RuntimeError):
    pass


def _plan_dict(info):
    d = {k: getattr(info, k) for k, _ in PlanInfo._fields_ if k not in ("region_words", "block_bytes")}
    d["region_words"] = list(info.region_words)[:info.tables]
    d["block_bytes"] = list(info.block_bytes)[:info.tables]
    return d


def plan_preview(world, rows_per_rank, left_range, right_range):
    """The regions-on-the-wire plan `world` ranks would make for tables of rows_per_rank[t] rows per rank and these global key
    ranges (mdb_dist_plan_preview: pure host computation, no GPU) -> dict as DistCtx.last_plan(), or None when that path would
    not serve the shape"""
    lib = load_library()
    _bind(lib)
    info = PlanInfo()
    rows = (c_uint64 * len(rows_per_rank))(*[int(r) for r in rows_per_rank])
    la, ra = (ctypes.c_int64 * 2)(int(left_range[0]), int(left_range[1])), (ctypes.c_int64 * 2)(int(right_range[0]), int(right_range[1]))
    rc = lib.mdb_dist_plan_preview(int(world), len(rows_per_rank), rows, la, ra, byref(info))
    if rc == 1:
        return None
    if rc:
        raise DistError(f"mdb_dist_plan_preview failed ({rc})")
    return _plan_dict(info)


class DatabaseDevice:
    """What DistCtx needs of a DeviceCtx, over the device context a database owns (mdb_database_device): lets a host build the
    exchange handle for a database itself - over RCCL with an id it ships, or over its own fabric - and hand it over with
    attach_to_database() (mdb_database_set_dist) instead of going through MIDORIDB_WORLD_SIZE / MIDORIDB_DIST_ID_FILE."""

    def __init__(self, db, device_index=0):
        from .dev import _bind as bind_dev
        bind_dev(db.lib)
        self.lib, self.h, self.device = db.lib, db.device_handle(), torch.device("cuda", device_index)

    def sync(self):
        if self.lib.mdb_dev_sync(self.h) != 0:
            raise DistError("mdb_dev_sync failed")


def _ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


class DistCtx:
    """One rank's handle on the sharded operator (mdb_dist*) for the GPU of a DeviceCtx."""

    def __init__(self, dev, world, rank, id_bytes=None, transport=None):
        self.lib = load_library()
        _bind(self.lib)
        self.dev = dev
        self.world = world
        self.rank = rank
        self._keep = transport          # the callbacks must outlive the handle
        h = c_void_p()
        if transport is not None:
            rc = self.lib.mdb_dist_init_transport(dev.h, world, rank, byref(transport), byref(h))
        else:
            if id_bytes is None or len(id_bytes) != MDB_DIST_ID_BYTES:
                raise DistError("a 128-byte communicator id is required")
            buf = (ctypes.c_char * MDB_DIST_ID_BYTES).from_buffer_copy(bytes(id_bytes))
            rc = self.lib.mdb_dist_init(dev.h, world, rank, buf, byref(h))
        if rc != 0:
            msg = self.lib.mdb_dev_last_error(dev.h)
            raise DistError(f"mdb_dist_init failed ({rc}): {msg.decode() if msg else ''}")
        self.h = h

    # -- rendezvous --------------------------------------------------------------------------------------------
    @staticmethod
    def unique_id():
        lib = load_library()
        _bind(lib)
        buf = ctypes.create_string_buffer(MDB_DIST_ID_BYTES)
        if lib.mdb_dist_unique_id(buf) != 0:
            raise DistError("mdb_dist_unique_id failed")
        return buf.raw

    @classmethod
    def from_torch(cls, dev, group=None):
        """The id travels through an initialised torch.distributed process group (any backend)."""
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(dev, world, rank, box[0])

    @classmethod
    def from_file(cls, dev, world, rank, path, timeout_s=120.0):
        lib = load_library()
        _bind(lib)
        buf = ctypes.create_string_buffer(MDB_DIST_ID_BYTES)
        if lib.mdb_dist_id_via_file(path.encode(), world, rank, timeout_s, buf) != 0:
            raise DistError(f"no communicator id through {path}")
        return cls(dev, world, rank, buf.raw)

    @classmethod
    def with_transport(cls, dev, world, rank, counts, alltoallv, allreduce):
        """counts(send: list[int] of world*n, n) -> list[int]; alltoallv(d_send, sendcounts, sdispls, d_recv, recvcounts,
        rdispls, elem_bytes, stream) with raw device pointers as ints; allreduce(vals: list[int]) -> list[int]."""
        def c_counts(_self, send, recv, n):
            try:
                out = counts([int(send[i]) for i in range(world * n)], n)
                for i, v in enumerate(out):
                    recv[i] = int(v)
                return 0
            except Exception as e:  # pragma: no cover - reported through the return code
                print("transport.counts failed:", e, flush=True)
                return -2

        def c_a2av(_self, d_send, sc, sd, d_recv, rc, rd, es, stream):
            try:
                alltoallv(d_send, [int(sc[i]) for i in range(world)], [int(sd[i]) for i in range(world)], d_recv,
                          [int(rc[i]) for i in range(world)], [int(rd[i]) for i in range(world)], int(es), stream)
                return 0
            except Exception as e:  # pragma: no cover
                print("transport.alltoallv failed:", e, flush=True)
                return -2

        def c_allred(_self, vals, n):
            try:
                out = allreduce([int(vals[i]) for i in range(n)])
                for i, v in enumerate(out):
                    vals[i] = int(v)
                return 0
            except Exception as e:  # pragma: no cover
                print("transport.allreduce failed:", e, flush=True)
                return -2

        t = Transport(None, _COUNTS(c_counts), _A2AV(c_a2av), _ALLRED(c_allred), _DESTROY(lambda _s: None))
        return cls(dev, world, rank, transport=t)

    @classmethod
    def over_host_group(cls, dev, group=None):
        """A bring-up / test transport: the blocks cross the ranks through HOST memory, carried by a torch.distributed process group of a
        CPU backend (gloo) - so N ranks can share ONE GPU (the GPU suite's world 2 / 4 / 8 runs, `bench.py --transport test`).  Every
        kernel, plan and count exchange is the product's; only the wire differs (timings of the wire mean nothing)."""
        import numpy as np
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)

        def counts(send, n):
            cin = torch.tensor(send, dtype=torch.int64)
            cout = torch.empty(world * n, dtype=torch.int64)
            dist.all_to_all_single(cout, cin, group=group)
            return cout.tolist()

        def alltoallv(d_send, sc, sd, d_recv, rc, rd, es, _stream):
            # any counts / displacements (the region exchange sends every peer the SAME counter array: displacement 0 for all)
            torch.cuda.synchronize()
            dev.sync()
            hi = max([sd[i] + sc[i] for i in range(world)] + [0])
            hs = np.zeros(max(hi, 1) * es, dtype=np.uint8)
            if hi and dev.lib.mdb_dev_d2h(dev.h, hs.ctypes.data, d_send, hi * es) != 0:
                raise DistError("host transport: device -> host copy failed")
            send = np.concatenate([hs[sd[i] * es:(sd[i] + sc[i]) * es] for i in range(world)]) if hi else np.zeros(0, dtype=np.uint8)
            nrecv = sum(rc)
            hr = torch.empty(max(nrecv, 1) * es, dtype=torch.uint8)
            dist.all_to_all_single(hr[:nrecv * es], torch.from_numpy(np.ascontiguousarray(send)), [c * es for c in rc], [c * es for c in sc],
                                   group=group)
            off = 0
            for i in range(world):
                if rc[i]:
                    piece = hr[off:off + rc[i] * es].numpy()
                    if dev.lib.mdb_dev_h2d(dev.h, d_recv + rd[i] * es, piece.ctypes.data, rc[i] * es) != 0:
                        raise DistError("host transport: host -> device copy failed")
                off += rc[i] * es

        def allreduce(vals):
            t = torch.tensor(vals, dtype=torch.int64)
            dist.all_reduce(t, group=group)
            return t.tolist()

        from .dev import _bind as bind_dev
        bind_dev(dev.lib)
        return cls.with_transport(dev, world, rank, counts, alltoallv, allreduce)

    # -- operator ----------------------------------------------------------------------------------------------
    def _chk(self, rc, what):
        if rc != 0:
            msg = self.lib.mdb_dist_last_error(self.h)
            raise DistError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def set_wire(self, mode):
        self._chk(self.lib.mdb_dist_set_wire(self.h, int(mode)), "set_wire")

    def last_wire32(self):
        return bool(self.lib.mdb_dist_last_wire32(self.h))

    def last_pruned(self):
        return bool(self.lib.mdb_dist_last_pruned(self.h))

    def last_fused(self):
        """the last join_group_count shipped first-level partition regions (mdb_dev_shard.hip) instead of keys"""
        return bool(self.lib.mdb_dist_last_fused(self.h))

    def last_plan(self):
        """-> dict of what the last regions-on-the-wire call planned (mdb_dist_last_plan), or None before the first such call"""
        info = PlanInfo()
        rc = self.lib.mdb_dist_last_plan(self.h, byref(info))
        if rc == 1:
            return None
        self._chk(rc, "last_plan")
        return _plan_dict(info)

    def set_phase_timing(self, on=True):
        self._chk(self.lib.mdb_dist_set_phase_timing(self.h, 1 if on else 0), "set_phase_timing")

    def last_phases(self):
        """-> {first_level_ms, wire_wait_ms, receiver_ms, device_ms} of the last regions-on-the-wire call (set_phase_timing)"""
        ms = (c_double * DIST_PHASES)()
        self._chk(self.lib.mdb_dist_last_phases(self.h, ms), "last_phases")
        return {"first_level_ms": ms[0], "wire_wait_ms": ms[1], "receiver_ms": ms[2], "device_ms": ms[3]}

    def set_key_ranges(self, left=None, right=None):
        """catalog statistics for WIRE_32 / WIRE_64 calls: the GLOBAL (smallest, largest) key of the left and right table (the same
        on every rank); rows outside the other table's range stay home, a key outside its own promised range is an error.
        None, None forgets them."""
        if left is None or right is None:
            self._chk(self.lib.mdb_dist_set_key_ranges(self.h, None, None), "set_key_ranges")
            return
        la, ra = (ctypes.c_int64 * 2)(int(left[0]), int(left[1])), (ctypes.c_int64 * 2)(int(right[0]), int(right[1]))
        self._chk(self.lib.mdb_dist_set_key_ranges(self.h, la, ra), "set_key_ranges")

    def last_received_left(self):
        return int(self.lib.mdb_dist_last_received_left(self.h))

    def join_group_count(self, keys_l, null_l, keys_r, null_r, out=None):
        """-> (keys[G], counts[G], joined rows on this rank): the groups whose key hashes to this rank."""
        n_l, n_r = keys_l.numel(), keys_r.numel()
        if out is None:
            cap = int(n_l * 1.3) + 4096 if self.world > 1 else max(n_l, 1)
            out = (torch.empty(cap, dtype=torch.int64, device=self.dev.device), torch.empty(cap, dtype=torch.int64, device=self.dev.device))
        ok, oc = out[0], out[1]
        g, j = c_uint64(), c_uint64()
        self._chk(self.lib.mdb_dist_join_group_count(self.h, _ptr(keys_l), _ptr(null_l), n_l, _ptr(keys_r), _ptr(null_r), n_r, _ptr(ok),
                                                     _ptr(oc), min(ok.numel(), oc.numel()), byref(g), byref(j)), "dist join_group_count")
        return ok[:g.value], oc[:g.value], j.value

    def _cols(self, cols):
        arr = (DistCol * max(len(cols), 1))()
        for i, c in enumerate(cols):
            v, nb, rid = (c + (None, None))[:3] if isinstance(c, tuple) else (c, None, None)
            arr[i] = DistCol(v.data_ptr(), nb.data_ptr() if nb is not None else None, rid.data_ptr() if rid is not None else None)
        return arr

    def _adopt(self, ptr, n, dtype):
        """a device buffer the library allocated (mdb_dev_alloc) as a torch tensor that releases it with mdb_dev_free"""
        return self.dev._adopt(c_void_p(int(ptr)), n, dtype)

    def shuffle_rows(self, keys, key_nulls, cols, flags=0):
        """cols: list of tensors or (values, nullbits, rid) tuples -> (list of (values, nullbits-or-None), received rows);
        every row travels to the rank its key hashes to (mdb_dist_shuffle_rows)."""
        nc = len(cols)
        ov, on = (c_void_p * max(nc, 1))(), (c_void_p * max(nc, 1))()
        got = c_uint64()
        self._chk(self.lib.mdb_dist_shuffle_rows(self.h, _ptr(keys), _ptr(key_nulls), keys.numel(), flags, self._cols(cols), nc, ov, on,
                                                 byref(got)), "dist shuffle_rows")
        n = got.value
        out = []
        for c in range(nc):
            src = cols[c][0] if isinstance(cols[c], tuple) else cols[c]
            out.append((self._adopt(ov[c], n, src.dtype), self._adopt(on[c], (n + 63) // 64, torch.int64) if on[c] else None))
        return out, n

    def broadcast_rows(self, n, cols):
        """every rank's n rows of `cols` to every rank -> (list of (values, nullbits-or-None), total rows) (mdb_dist_broadcast_rows)"""
        nc = len(cols)
        ov, on = (c_void_p * nc)(), (c_void_p * nc)()
        got = c_uint64()
        self._chk(self.lib.mdb_dist_broadcast_rows(self.h, n, self._cols(cols), nc, ov, on, byref(got)), "dist broadcast_rows")
        m = got.value
        out = []
        for c in range(nc):
            src = cols[c][0] if isinstance(cols[c], tuple) else cols[c]
            out.append((self._adopt(ov[c], m, src.dtype), self._adopt(on[c], (m + 63) // 64, torch.int64) if on[c] else None))
        return out, m

    def join_pairs(self, keys_l, null_l, cols_l, keys_r, null_r, cols_r):
        """-> (key[J], [(values, nullbits) of cols_l], [... of cols_r], J): the joined rows whose key hashes to this rank."""
        nl, nr = len(cols_l), len(cols_r)
        ok = c_void_p()
        ol, oln = (c_void_p * max(nl, 1))(), (c_void_p * max(nl, 1))()
        orr, orn = (c_void_p * max(nr, 1))(), (c_void_p * max(nr, 1))()
        rows = c_uint64()
        self._chk(self.lib.mdb_dist_join_pairs(self.h, _ptr(keys_l), _ptr(null_l), keys_l.numel(), self._cols(cols_l), nl, _ptr(keys_r),
                                               _ptr(null_r), keys_r.numel(), self._cols(cols_r), nr, byref(ok), ol, oln, orr, orn, byref(rows)),
                  "dist join_pairs")
        J = rows.value

        def side(cols, ov, on):
            out = []
            for c in range(len(cols)):
                src = cols[c][0] if isinstance(cols[c], tuple) else cols[c]
                out.append((self._adopt(ov[c], J, src.dtype), self._adopt(on[c], (J + 63) // 64, torch.int64) if on[c] else None))
            return out
        return self._adopt(ok.value, J, torch.int64), side(cols_l, ol, oln), side(cols_r, orr, orn), J

    def attach_to_database(self, db):
        """the database owns the handle from now on (destroyed by database_close); this object keeps the transport callbacks alive"""
        db.set_dist(self.h)
        db._dist_keepalive = self
        self.h = None

    def join_group_count_multi(self, keys_l, null_l, rights):
        """rights = [(keys, nullbits or None), ...] (2 or 3 tables, all joined to keys_l on one key) -> (keys[G], counts[G], joined rows on this
        rank), or None when the shape is not served and the caller has to chain two-table calls"""
        nr = len(rights)
        kr = (c_void_p * nr)(*[r[0].data_ptr() for r in rights])
        nb = (c_void_p * nr)(*[(r[1].data_ptr() if r[1] is not None else None) for r in rights])
        ns = (c_uint64 * nr)(*[r[0].numel() for r in rights])
        ok, oc, g, j = c_void_p(), c_void_p(), c_uint64(), c_uint64()
        rc = self.lib.mdb_dist_join_group_count_multi_alloc(self.h, _ptr(keys_l), _ptr(null_l), keys_l.numel(), nr, kr, nb, ns, byref(ok), byref(oc),
                                                            byref(g), byref(j))
        if rc == 1:
            return None
        self._chk(rc, "dist join_group_count_multi")
        return self._adopt(ok.value, g.value, torch.int64), self._adopt(oc.value, g.value, torch.int64), j.value

    def group_count_keys(self, keys, nulls=None):
        """GROUP BY + COUNT(*) of one sharded key column -> (keys[G], counts[G]) of the keys that live on this rank, in unspecified
        order; None when the form does not serve the column (every rank gets None: exchange the rows and group locally)"""
        ok, oc, g = c_void_p(), c_void_p(), c_uint64()
        rc = self.lib.mdb_dist_group_count_keys_alloc(self.h, _ptr(keys), _ptr(nulls), keys.numel(), byref(ok), byref(oc), byref(g))
        if rc == 1:
            return None
        self._chk(rc, "dist group_count_keys")
        return self._adopt(ok.value, g.value, torch.int64), self._adopt(oc.value, g.value, torch.int64)

    def allreduce_sum(self, vals):
        arr = (c_uint64 * len(vals))(*[int(v) for v in vals])
        self._chk(self.lib.mdb_dist_allreduce_sum_u64(self.h, arr, len(vals)), "allreduce")
        return [int(v) for v in arr]

    def barrier(self):
        self._chk(self.lib.mdb_dist_barrier(self.h), "barrier")

    def close(self):
        if getattr(self, "h", None):
            self.lib.mdb_dist_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

