"""ctypes loader for libmidoridb_amd.so (no fallback: a missing library is an error)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def library_path():
    """MDB_LIBRARY overrides the path (the sanitizer build `make -C midoridb_amd/csrc asan`, tests/test_sanitizers.py)."""
    return os.environ.get("MDB_LIBRARY") or os.path.join(_HERE, "libmidoridb_amd.so")


def load_library():
    """Load the C-ABI library once.  Raises if it has not been built (``__graft_entry__.build()``)."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C midoridb_amd/csrc).  There is no CPU fallback for the device path."
            )
        _LIB = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        _follow_environment(_LIB)
    return _LIB


def _follow_environment(lib):
    """The library reads its MDB_* knobs once per process and keeps them (mdb_knob, mdb_dev_core.hip); tests and same-process A/B scripts flip
    knobs through os.environ / monkeypatch while the process runs: every change of an MDB_* variable made through Python drops what the
    library has kept (mdb_dev_reload_knobs)."""
    lib.mdb_dev_reload_knobs.restype = None
    lib.mdb_dev_reload_knobs.argtypes = []
    if getattr(os, "_mdb_knobs_followed", False):
        return
    real_put, real_unset = os.putenv, os.unsetenv

    def _name(key):
        return key.decode("ascii", "replace") if isinstance(key, bytes) else str(key)

    def putenv(key, value):
        real_put(key, value)
        if _name(key).startswith("MDB_"):
            lib.mdb_dev_reload_knobs()

    def unsetenv(key):
        real_unset(key)
        if _name(key).startswith("MDB_"):
            lib.mdb_dev_reload_knobs()

    os.putenv, os.unsetenv = putenv, unsetenv     # (os.environ's __setitem__ / __delitem__ call these module-level functions)
    os._mdb_knobs_followed = True
