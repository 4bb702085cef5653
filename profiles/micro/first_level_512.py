"""What the first-level scatter of the right table costs at 512 digits, by key window and call shape (10^8 x 10^8 rows):
the benchmark's variant D (23-bit R-based window, right table first, min-max recorded, one level) against tables whose keys
both lie in the same 23-bit window (one level, left table first, no min-max) and against variant U (27-bit window, two levels)."""
import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
n = 100_000_000
cases = {
    "D (left 27 bits, right lowest sixteenth)": (dev.gen_keys(n, 0, n, 42, 0), dev.gen_keys(n, 0, n, 43, n // 16)),
    "both tables dup16 inside 23 bits": (dev.gen_keys(n, 0, n, 42, n // 16), dev.gen_keys(n, 0, n, 43, n // 16)),
    "U (both 27 bits, unique)": (dev.gen_keys(n, 0, n, 42, 0), dev.gen_keys(n, 0, n, 43, 0)),
}
for name, (kl, kr) in cases.items():
    for _ in range(3):
        k, c, f, j = dev.join_group_count(kl, None, kr, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        k, c, f, j = dev.join_group_count(kl, None, kr, None)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    dev.prof_enable(True); dev.prof_reset(); dev.join_group_count(kl, None, kr, None); prof = dev.prof_read(); dev.prof_enable(False)
    print(name, "| form", dev.last_join_form(), "levels", dev.last_join_levels(), "filter", dev.last_join_filter(), "ms %.3f" % ms, "groups", k.numel(),
          {k2: round(v[1], 3) for k2, v in prof.items() if v[1] > 0.03}, flush=True)
    del k, c, f
