"""Row-order join's leaf: 8 lanes per piece against 4 (MDB_RJ_LPP=4: twice the pieces per instruction, the pieces' words 4 .. 7 in a second tier), same box,
alternating: 10^8 x 10^8 unique keys, one and two payload cells.    python profiles/micro/rj_lpp_ab.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
n = 100_000_000
a, b = dev.gen_keys(n, 0, n, 42, 0), dev.gen_keys(n, 0, n, 43, 0)
for cells in (1, 2):
    pay = [b * 3 + 1, b * 5 - 2][:cells]
    for rep in range(2):
        for knob in ("8", "4"):
            os.environ["MDB_RJ_LPP"] = knob
            for _ in range(2):
                got = dev.join_payload(a, None, b, None, pay)
            assert got is not None and torch.equal(got[0], a * 3 + 1) and (cells == 1 or torch.equal(got[1], a * 5 - 2))
            dev.prof_enable(True)
            dev.prof_reset()
            for _ in range(3):
                dev.join_payload(a, None, b, None, pay)
            kern = {k: round(v[1] / 3, 4) for k, v in dev.prof_read().items() if v[0] > 0 and k.startswith("rowjoin")}
            dev.prof_enable(False)
            print(json.dumps({"cells": cells, "MDB_RJ_LPP": knob, **kern}), flush=True)
