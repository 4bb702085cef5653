import sys, time, json, os
sys.path.insert(0, os.getcwd())
import torch
from midoridb_amd.dev import DeviceCtx
from midoridb_amd import dev as D
dev = DeviceCtx()
n = 100_000_000
i = torch.arange(n, dtype=torch.int64, device=dev.device)
k2 = (i * 7919) % 512
k3 = (i * 104729) % 300
keys = [(k2, None, None, D.T_INT64, False), (k3, None, None, D.T_INT64, False)]
def run():
    return dev.group_count_multi(keys, n)
for knob in ("1", "0", "1"):
    os.environ["MDB_GROUP_MULTI_PACKED"] = knob
    f, c = run(); torch.cuda.synchronize()
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); f, c = run(); torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
    print("packed", knob, "groups", f.numel(), "sum", int(c.sum()), "ms", [round(x, 3) for x in t])
# random two columns with NULL-free values and 10^6 combinations
g = torch.Generator(device=dev.device); g.manual_seed(1)
a = torch.randint(0, 1000, (n,), device=dev.device, generator=g, dtype=torch.int64)
b = torch.randint(-500, 500, (n,), device=dev.device, generator=g, dtype=torch.int64)
keys = [(a, None, None, D.T_INT64, False), (b, None, None, D.T_INT64, False)]
res = {}
for knob in ("1", "0"):
    os.environ["MDB_GROUP_MULTI_PACKED"] = knob
    f, c = run(); torch.cuda.synchronize()
    t = []
    for _ in range(3):
        t0 = time.perf_counter(); f, c = run(); torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
    res[knob] = (f.clone(), c.clone())
    print("random 10^6 combos packed", knob, "groups", f.numel(), "ms", [round(x, 3) for x in t])
print("same result:", torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][1], res["0"][1]))
# kernels of the packed form
os.environ["MDB_GROUP_MULTI_PACKED"] = "1"
keys = [(k2, None, None, D.T_INT64, False), (k3, None, None, D.T_INT64, False)]
dev.prof_enable(True); dev.prof_reset(); run(); prof = dev.prof_read(); dev.prof_enable(False)
print({k: round(v[1], 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:10]})
# few combinations: 16 x 50
a = torch.randint(0, 16, (n,), device=dev.device, generator=g, dtype=torch.int64)
b = torch.randint(100, 150, (n,), device=dev.device, generator=g, dtype=torch.int64)
keys = [(a, None, None, D.T_INT64, False), (b, None, None, D.T_INT64, False)]
for fused in ("1", "0"):
    os.environ["MDB_GROUP_MULTI_FUSED"] = fused
    f, c = run(); torch.cuda.synchronize()
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); f, c = run(); torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
    print("16 x 50 combinations, fused", fused, "groups", f.numel(), "sum", int(c.sum()), "ms", [round(x, 3) for x in t])
