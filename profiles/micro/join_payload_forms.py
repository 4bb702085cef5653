"""mdb_dev_join_payload, the row-order form (mdb_dev_rowjoin.hip, MDB_ROWJOIN=2) against the older forms (MDB_ROWJOIN=0): unique keys on both
sides (two permutations of 0..n-1), one and two payload cells, by table size:
    python profiles/micro/join_payload_forms.py [rows ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from midoridb_amd.dev import DeviceCtx

sizes = [int(float(a)) for a in sys.argv[1:]] or [10_000_000, 30_000_000, 100_000_000]
dev = DeviceCtx(0)
out = []
for n in sizes:
    a = dev.gen_keys(n, 0, n, 42, 0)
    b = dev.gen_keys(n, 0, n, 43, 0)
    pay = [torch.arange(n, dtype=torch.int64, device=dev.device), torch.arange(n, dtype=torch.int64, device=dev.device) * 3]
    for cells in (1, 2):
        row = {"rows": n, "cells": cells}
        for form in ("0", "2"):
            os.environ["MDB_ROWJOIN"] = form
            for _ in range(3):
                r = dev.join_payload(a, None, b, None, pay[:cells])
            assert r is not None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                r = dev.join_payload(a, None, b, None, pay[:cells])
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            dev.prof_enable(True)
            dev.prof_reset()
            dev.join_payload(a, None, b, None, pay[:cells])
            kern = {k: round(v[1], 4) for k, v in dev.prof_read().items()}
            dev.prof_enable(False)
            row["row_order_ms" if form == "2" else "older_forms_ms"] = round(ms, 4)
            row["row_order_kernels" if form == "2" else "older_forms_kernels"] = kern
        out.append(row)
        print(json.dumps(row), flush=True)
    del a, b, pay
