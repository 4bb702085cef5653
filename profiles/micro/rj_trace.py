"""Row-order join's leaf: where a digit's time goes (MDB_RJ_TRACE: one workgroup's wave 0 stamps its phase boundaries).   python profiles/micro/rj_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
n = 100_000_000
a, b = dev.gen_keys(n, 0, n, 42, 0), dev.gen_keys(n, 0, n, 43, 0)
pay = [b * 3 + 1]
for _ in range(2):
    dev.join_payload(a, None, b, None, pay)
os.environ["MDB_RJ_TRACE"] = "1"
dev.join_payload(a, None, b, None, pay)
