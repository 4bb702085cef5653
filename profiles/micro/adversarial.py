"""Adversarial key distributions through the fused join + GROUP BY operator: correctness vs closed forms, and time."""
import sys, time, torch
sys.path.insert(0, '.')
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
i = torch.arange(n, dtype=torch.int64, device=dev.device)
cases = {
    "all rows one key": (torch.full((n,), 7, dtype=torch.int64, device=dev.device), torch.full((n // 10,), 7, dtype=torch.int64, device=dev.device)),
    "two keys": (i % 2, i[: n // 10] % 2),
    "sorted unique": (i, i),
    "multiples of 2^20": (i << 20, i << 20),
    "negative sorted": (-i, -i),
    "90% one key + unique rest": (torch.where(i % 10 != 0, torch.zeros_like(i), i), torch.where(i % 10 != 0, torch.zeros_like(i), i)[: n // 4]),
}
for name, (a, b) in cases.items():
    a, b = a.contiguous(), b.contiguous()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    k, c, f, j = dev.join_group_count(a, None, b, None)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    # closed-form checks: sum of counts = joined rows = sum over keys cl*cr
    ua, ca = torch.unique(a, return_counts=True)
    ub, cb = torch.unique(b, return_counts=True)
    common = torch.isin(ua, ub)
    idx = torch.searchsorted(ub, ua[common])
    want_j = int((ca[common] * cb[idx]).sum())
    ok = j == want_j and k.numel() == int(common.sum()) and int(c.sum()) == want_j and bool((a[f.long()] == k).all())
    print(f"{name:28s} {ms:9.2f} ms  groups {k.numel():9d}  joined {j:16d}  {'OK' if ok else 'MISMATCH'}")
