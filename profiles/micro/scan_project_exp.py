"""Timing of the single-pass scan + WHERE + projection kernel, 10^8 rows, 50 % selectivity (the 8192-row workgroup variant
that was measured with it - MDB_SP_SPANS=4, 0.36 vs 0.33 ms - has been removed from the source)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from midoridb_amd import dev as D
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
n = 100_000_000
v = dev.gen_keys(n, 0, n, 7, 0)
w = dev.gen_keys(n, 0, n, 8, 0)
prog = [(D.P_CMP_COL_CONST, D.CMP_GT, D.T_INT64, 0, 0, n // 2)]
for name, proj in (("same column", [(v, None)]), ("two other columns", [(w, None), (v, None)])):
    for _ in range(2):
        dev.filter_project(prog, [(v, None, None)], n, proj)
    dev.prof_enable(True); dev.prof_reset()
    for _ in range(5):
        m, _o = dev.filter_project(prog, [(v, None, None)], n, proj)
    prof = dev.prof_read(); dev.prof_enable(False)
    print(os.environ.get("MDB_SP_SPANS", "8"), name, m, {k: round(t / 5, 4) for k, (c, t) in prof.items()})
