"""Micro-benchmark of mdb_dev_partition_by_dest (the multi-GPU send-side partition) at 10^8 keys for 1..8 destinations."""
import sys, time, torch
sys.path.insert(0, '.')
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
n = 100_000_000
a = dev.gen_keys(n, 0, n, 42, 0)
out = torch.empty(n, dtype=torch.int64, device=dev.device)
K32 = "--keys32" in sys.argv
out = out.view(torch.int32) if K32 else out
for nd in (1, 2, 4, 8):
    for with_rid in (False, True):
        for _ in range(2):
            dev.partition_by_dest(a, None, nd, out=out, with_rid=with_rid, keys32=K32)
        dev.prof_enable(True); dev.prof_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            r = dev.partition_by_dest(a, None, nd, out=out, with_rid=with_rid, keys32=K32)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        prof = dev.prof_read(); dev.prof_enable(False)
        print(f"n_dest={nd} rid={with_rid}: {ms:.3f} ms", {k: round(v[1] / 5, 3) for k, v in prof.items() if v[1] > 0.05}, "counts", r[1][:3])
