"""k_probe_stage with parts switched off (MDB_PROBE_ABLATE: 1 no table lookups, 2 no staging writes, 4 nothing listed): where its time goes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from midoridb_amd.dev import DeviceCtx
n = 100_000_000
dev = DeviceCtx(0)
a = dev.gen_keys(n, 0, n, 42, 0)
b = dev.gen_keys(n, 0, n, 43, n // 16)
sa = dev.key_range(a) + (1,)
sb = dev.key_range(b) + (0,)
os.environ["MDB_PROBE"] = "1"
for ab in ("0", "1", "2", "3", "4", "0"):
    os.environ["MDB_PROBE_ABLATE"] = ab
    dev.call_stats(a, sa, b, sb)
    try:
        for _ in range(2):
            dev.join_group_count(a, None, b, None, want_first=False)
        dev.prof_enable(True); dev.prof_reset()
        for _ in range(5):
            dev.join_group_count(a, None, b, None, want_first=False)
        prof = {kk: round(v[1] / 5, 4) for kk, v in dev.prof_read().items() if kk.startswith("probe") or kk.startswith("part_scatter_l0_w32")}
        dev.prof_enable(False)
        print("ablate", ab, prof, flush=True)
    except Exception as e:
        print("ablate", ab, "error", e, flush=True)
    dev.call_stats()
