"""midoridb_amd - MI355X (gfx950) execution path for MidoriDB's SELECT executor.

The product is the C-ABI shared library ``libmidoridb_amd.so`` (host C + hand-written HIP,
built from ``midoridb_amd/csrc``).  This package is only the Python binding used by the tests
and ``bench.py``: ``midoridb_amd.lib`` loads the library (and fails loudly when it is
missing - there is no CPU fallback), ``midoridb_amd.dev`` wraps the device operators of
``include/mdb_dev.h`` for torch tensors, ``midoridb_amd.query`` mirrors the reference's
``query_execute()/query_cur_step()/query_column_int64()`` API (``include/mdb_query.h``).
"""
from .lib import load_library, library_path  # noqa: F401
