"""GROUP BY + COUNT(*) over 10^8 rows as a function of the number of distinct values (values spread over a wide range,
so the direct small-range path does not apply, and the same inside a small range): time and the kernels that ran."""
import sys, time, torch
sys.path.insert(0, '.')
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
n = 100_000_000
base = dev.gen_keys(n + 64 * 4096, 0, n + 64 * 4096, 42, 0)
case = 0
for D in (10, 1000, 10_000, 100_000, 1_000_000, 10_000_000):
    for wide in (True, False):
        case += 1       # (every case its own length: what the operator learned about a column is remembered by address and length)
        k = base[: n + 4096 * case] % D
        if wide:
            k = k * 1_000_003 + 7          # spread: span = D * 1e6
        for _ in range(3):
            first, cnt = dev.group_count(k, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            first, cnt = dev.group_count(k, None)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        dev.prof_enable(True); dev.prof_reset()
        first, cnt = dev.group_count(k, None)
        prof = {kk: round(v[1], 2) for kk, v in dev.prof_read().items() if v[1] > 0.2}
        dev.prof_enable(False)
        print(f"D={D:>9} {'wide ' if wide else 'dense'} {ms:8.3f} ms  groups {first.numel():>9}  {prof}")
        del k
