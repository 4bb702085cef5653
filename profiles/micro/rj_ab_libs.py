"""Row-order join, two builds of the library on one box, alternating (each in its own process): 10^8 x 10^8 unique keys, one / two payload cells.
    python profiles/micro/rj_ab_libs.py <libA.so> <libB.so> [rounds]"""
import json, os, subprocess, sys
CHILD = r'''
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
n = 100_000_000
a, b = dev.gen_keys(n, 0, n, 42, 0), dev.gen_keys(n, 0, n, 43, 0)
for cells in (1, 2):
    pay = [b * 3 + 1, b * 5 - 2][:cells]
    for _ in range(2):
        got = dev.join_payload(a, None, b, None, pay)
    assert got is not None and torch.equal(got[0], a * 3 + 1) and (cells == 1 or torch.equal(got[1], a * 5 - 2))
    dev.prof_enable(True); dev.prof_reset()
    for _ in range(3):
        dev.join_payload(a, None, b, None, pay)
    kern = {k: round(v[1] / 3, 4) for k, v in dev.prof_read().items() if v[0] > 0 and k.startswith("rowjoin")}
    dev.prof_enable(False)
    print(json.dumps({"lib": os.path.basename(os.environ["MDB_LIBRARY"]), "cells": cells, **kern}), flush=True)
'''
libs, rounds = sys.argv[1:3], int(sys.argv[3]) if len(sys.argv) > 3 else 2
for _ in range(rounds):
    for lib in libs:
        subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MDB_LIBRARY=os.path.abspath(lib)), check=False)
