"""Row-order join: does the leaf's time follow the number of (tile, digit) pieces?  10^8 left rows against a right table of 2^26 unique keys
(4096 digits: pieces of 8 words) and against one of 10^8 unique keys in a window of 2^27 values (8192 digits: pieces of 4).
    python profiles/micro/rj_piece_length.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
n = 100_000_000
for nr in (1 << 26, n):
    b = dev.gen_keys(nr, 0, nr, 43, 0)
    a = dev.gen_keys(n, 0, n, 42, 0) % nr if nr < n else dev.gen_keys(n, 0, n, 42, 0)
    pay = [b * 3 + 1]
    for _ in range(2):
        got = dev.join_payload(a, None, b, None, pay)
    assert got is not None and torch.equal(got[0], a * 3 + 1)
    dev.prof_enable(True)
    dev.prof_reset()
    dev.join_payload(a, None, b, None, pay)
    kern = {k: round(v[1], 4) for k, v in dev.prof_read().items() if v[0] > 0}
    dev.prof_enable(False)
    print(json.dumps({"left_rows": n, "right_rows": nr, "plan_key_bits": dev.last_plan()["key_bits"], **kern}), flush=True)
    del a, b, pay, got
