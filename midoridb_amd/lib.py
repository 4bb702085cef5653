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
    return _LIB
