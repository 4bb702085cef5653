"""k_bg_band_sort's parts by ablation (MDB_BG_ABLATE: 1 no cursor atomics, 2 no words written, 4 words written in tile order): kernel times only -
the results of an ablated run are garbage (the caller's other forms answer).   python profiles/micro/group_banded_ablate.py [rows]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from midoridb_amd.dev import DeviceCtx
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
dev = DeviceCtx(0)
keys = dev.gen_keys(n, 0, n, 43, n // 16)
for ab in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("0", "1", "2", "3", "4", "5")):
    os.environ["MDB_BG_ABLATE"] = ab
    for _ in range(2):
        dev.group_count(keys, None)
    dev.prof_enable(True)
    dev.prof_reset()
    dev.group_count(keys, None)
    kern = {k: round(v[1], 4) for k, v in dev.prof_read().items() if k.startswith("group_band") or k == "scan_small"}
    dev.prof_enable(False)
    print(json.dumps({"ablate": ab, **kern}), flush=True)
