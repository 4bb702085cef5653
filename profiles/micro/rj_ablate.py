"""The row-order join's leaf with phases left out (MDB_RJ_ABLATE: 1 no build, 2 no probe, 4 no cell stores, 8 no right cells read, 16 pieces longer than the lanes of a piece are cut off - wrong results: what the rare long pieces cost): kernel times
at 10^8 x 10^8 unique keys, one cell.   python profiles/micro/rj_ablate.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from midoridb_amd.dev import DeviceCtx
n = 100_000_000
dev = DeviceCtx(0)
a, b = dev.gen_keys(n, 0, n, 42, 0), dev.gen_keys(n, 0, n, 43, 0)
pay = [torch.arange(n, dtype=torch.int64, device=dev.device)]
for ab in ("0", "16", "17", "18", "1", "2", "4"):
    os.environ["MDB_RJ_ABLATE"] = ab
    a2, b2 = a.clone(), b.clone()     # (a call that fails - the ablations deliver wrong results - is remembered by the columns' addresses)
    if ab == "0":
        for _ in range(2):
            dev.join_payload(a2, None, b2, None, pay)
    dev.prof_enable(True)
    dev.prof_reset()
    dev.join_payload(a2, None, b2, None, pay)
    kern = {k: round(v[1], 4) for k, v in dev.prof_read().items() if v[0] > 0 and k.startswith("rowjoin")}
    dev.prof_enable(False)
    print(json.dumps({"ablate": int(ab), **kern}), flush=True)
    del a2, b2
