import sys, time, torch
sys.path.insert(0, "/root/repo")
from midoridb_amd import dev as D
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
n = 100_000_000
v = dev.gen_keys(n, 0, n, 7, 0)
for prog, name in (([(D.P_CMP_COL_CONST, D.CMP_GT, D.T_INT64, 0, 0, n // 2)], "one comparison"),
                   ([(D.P_CMP_COL_CONST, D.CMP_GT, D.T_INT64, 0, 0, n // 2), (D.P_CMP_COL_CONST, D.CMP_LT, D.T_INT64, 0, 0, n - 5), (D.P_AND, 0, 0, 0, 0, 0)], "range")):
    for _ in range(2):
        sel = dev.filter(prog, [(v, None, None)], n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        sel = dev.filter(prog, [(v, None, None)], n)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    dev.prof_enable(True); dev.prof_reset(); dev.filter(prog, [(v, None, None)], n); prof = dev.prof_read(); dev.prof_enable(False)
    print(name, "rows out", sel.numel(), "ms %.3f" % ms, {k: round(x[1], 3) for k, x in prof.items()}, flush=True)
