"""Scratch-arena size after the headline statement (10^8 x 10^8 rows through query_execute) and after the raw operators:
    python profiles/micro/arena_need.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from midoridb_amd.dev import DeviceCtx
dev = DeviceCtx(0)
dev.lib.mdb_dev_arena_bytes.restype = __import__("ctypes").c_size_t
n = 100_000_000
a, b = dev.gen_keys(n, 0, n, 42, 0), dev.gen_keys(n, 0, n, 43, n // 16)
dev.join_group_count(a, None, b, None)
print("join + GROUP BY (variant D): arena %.2f GB" % (dev.lib.mdb_dev_arena_bytes(dev.h) / 1e9))
dev.call_stats(a, (0, n - 1), b, (0, n // 16 - 1))
dev.join_group_count(a, None, b, None)
dev.call_stats()
print("... with catalog statistics: arena %.2f GB" % (dev.lib.mdb_dev_arena_bytes(dev.h) / 1e9))
dev.group_count(b, None)
print("GROUP BY of B: arena %.2f GB" % (dev.lib.mdb_dev_arena_bytes(dev.h) / 1e9))
