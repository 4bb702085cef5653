import sys, torch
sys.path.insert(0, '.')
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
n = 20_000_000
a = torch.full((n,), 7, dtype=torch.int64, device=dev.device)
b = torch.full((n // 10,), 7, dtype=torch.int64, device=dev.device)
dev.join_group_count(a, None, b, None)
dev.prof_enable(True); dev.prof_reset()
dev.join_group_count(a, None, b, None)
p = dev.prof_read()
for k, v in sorted(p.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{k:28s} launches {v[0]:3d}  {v[1]:9.3f} ms")
