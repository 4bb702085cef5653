"""What made `unordered.ms_per_step` 2.82 ms in profiles/r05/bench_driver_command.json (0.51 everywhere else)?
300 calls of the any-order operator over the headline's tables (variant D, 10^8 rows per table), preceded by the ordered operator's
calls like bench.py's sequence; every call's wall time with the context's counters (mdb_dev_counters) before and after it: a slow call
is printed with what it paid for (retry, key sample, arena growth, allocator miss)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from midoridb_amd.dev import DeviceCtx

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = DeviceCtx(0)
a = dev.gen_keys(n, 0, n, 42, 0)
b = dev.gen_keys(n, 0, n, 43, n // 16)
out = None
times = []
def one(fn, tag):
    c0 = dev.counters()
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) * 1e3
    c1 = dev.counters()
    d = {k: c1[k] - c0[k] for k in c0 if c1[k] != c0[k] and k != "operator_calls"}
    times.append((tag, ms, d, dev.arena_bytes() if hasattr(dev, "arena_bytes") else None))
for i in range(40):
    one(lambda: dev.join_group_count(a, None, b, None, want_first=False), "ordered")
for i in range(300):
    one(lambda: dev.join_group_count_unordered(a, None, b, None), "unordered")
for tag in ("ordered", "unordered"):
    v = sorted(ms for t, ms, d, ab in times if t == tag)
    print(f"{tag}: {len(v)} calls, min {v[0]:.3f} median {v[len(v) // 2]:.3f} p90 {v[9 * len(v) // 10]:.3f} max {v[-1]:.3f} ms")
med = {tag: sorted(ms for t, ms, d, ab in times if t == tag)[len([1 for t, *_ in times if t == tag]) // 2] for tag in ("ordered", "unordered")}
for i, (tag, ms, d, ab) in enumerate(times):
    if ms > 1.5 * med[tag] or d:
        print(f"call {i} ({tag}): {ms:.3f} ms, paid {d}, arena {ab}")
