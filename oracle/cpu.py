"""cpu.py - TEST INFRASTRUCTURE ONLY: ctypes access to oracle/liboracle.so (cpu_naive.c, cpu_hash.c)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("MDB_ORACLE_LIBRARY") or os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (done by __graft_entry__.build())")
        L = ctypes.CDLL(path)
        P, u64 = ctypes.c_void_p, ctypes.c_uint64
        L.orc_naive_join_pairs.argtypes = [P, P, u64, P, P, u64, P, P]
        L.orc_naive_join_pairs.restype = u64
        L.orc_naive_group_count.argtypes = [P, P, u64, P, P]
        L.orc_naive_group_count.restype = u64
        L.orc_naive_join_group_count.argtypes = [P, P, u64, P, P, u64, P, P, ctypes.POINTER(u64)]
        L.orc_naive_join_group_count.restype = u64
        L.orc_hash_join_group_count.argtypes = [P, P, u64, P, P, u64, ctypes.c_int, P, P, P, ctypes.POINTER(u64),
                                                ctypes.POINTER(u64)]
        L.orc_hash_join_group_count.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _prep(keys, nulls):
    k = np.ascontiguousarray(keys, dtype=np.int64)
    n = None if nulls is None else np.ascontiguousarray(nulls, dtype=np.uint8)
    return k, n


def naive_join_pairs(kl, nl, kr, nr):
    kl, nl = _prep(kl, nl)
    kr, nr = _prep(kr, nr)
    j = lib().orc_naive_join_pairs(_p(kl), _p(nl), len(kl), _p(kr), _p(nr), len(kr), None, None)
    ol = np.zeros(max(j, 1), dtype=np.uint32)
    orr = np.zeros(max(j, 1), dtype=np.uint32)
    lib().orc_naive_join_pairs(_p(kl), _p(nl), len(kl), _p(kr), _p(nr), len(kr), _p(ol), _p(orr))
    return ol[:j].astype(np.int64), orr[:j].astype(np.int64)


def naive_group_count(keys, nulls):
    k, n = _prep(keys, nulls)
    first = np.zeros(max(len(k), 1), dtype=np.uint32)
    cnt = np.zeros(max(len(k), 1), dtype=np.int64)
    g = lib().orc_naive_group_count(_p(k), _p(n), len(k), _p(first), _p(cnt))
    return first[:g].astype(np.int64), cnt[:g]


def naive_join_group_count(kl, nl, kr, nr):
    """The north-star query exactly as the reference runs it (nested loop + quadratic GROUP BY)."""
    kl, nl = _prep(kl, nl)
    kr, nr = _prep(kr, nr)
    cap = max(len(kl), 1)		# at most one group per distinct left key
    ok = np.zeros(cap, dtype=np.int64)
    oc = np.zeros(cap, dtype=np.int64)
    joined = ctypes.c_uint64()
    g = lib().orc_naive_join_group_count(_p(kl), _p(nl), len(kl), _p(kr), _p(nr), len(kr), _p(ok), _p(oc),
                                         ctypes.byref(joined))
    return ok[:g], oc[:g], joined.value


def hash_join_group_count(kl, nl, kr, nr, nthreads=8):
    kl, nl = _prep(kl, nl)
    kr, nr = _prep(kr, nr)
    cap = max(len(kl), 1)
    ok = np.zeros(cap, dtype=np.int64)
    oc = np.zeros(cap, dtype=np.int64)
    of = np.zeros(cap, dtype=np.uint32)
    g, j = ctypes.c_uint64(), ctypes.c_uint64()
    rc = lib().orc_hash_join_group_count(_p(kl), _p(nl), len(kl), _p(kr), _p(nr), len(kr), nthreads, _p(ok), _p(oc),
                                         _p(of), ctypes.byref(g), ctypes.byref(j))
    if rc != 0:
        raise MemoryError("orc_hash_join_group_count failed")
    return ok[:g.value], oc[:g.value], of[:g.value].astype(np.int64), j.value
