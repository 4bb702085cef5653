import sys, time, torch
sys.path.insert(0, "/root/repo")
from midoridb_amd import dev as D
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
n = 100_000_000
v = dev.gen_keys(n, 0, n, 7, 0)
w = dev.gen_keys(n, 0, n, 8, 0)
prog = [(D.P_CMP_COL_CONST, D.CMP_GT, D.T_INT64, 0, 0, n // 2), (D.P_CMP_COL_CONST, D.CMP_LT, D.T_INT64, 0, 0, n - 5), (D.P_AND, 0, 0, 0, 0, 0)]
for proj, name in (([(v, None)], "range, project the same column"), ([(v, None), (w, None)], "range, project two columns")):
    for _ in range(2):
        m, _o = dev.filter_project(prog, [(v, None, None)], n, proj)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        m, _o = dev.filter_project(prog, [(v, None, None)], n, proj)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    dev.prof_enable(True); dev.prof_reset(); dev.filter_project(prog, [(v, None, None)], n, proj); prof = dev.prof_read(); dev.prof_enable(False)
    print(name, m, "ms %.3f" % ms, {k: round(x[1], 3) for k, x in prof.items()}, flush=True)
