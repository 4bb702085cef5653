"""_exchange_model.py - TEST INFRASTRUCTURE ONLY.  A CPU model of the exchange protocol of midoridb_amd/csrc/mdb_dist.hip
(mdb_dist_shuffle_rows / mdb_dist_join_pairs / mdb_dist_join_group_count) over torch.distributed with the gloo backend, with
the oracle's functions where the product runs HIP kernels.

The product's exchange needs a GPU (its partition, gather and join steps are device kernels), so on a CPU-only box the
world_size-2 test (tests/test_distributed_gloo.py) cannot run it; what it CAN pin is the protocol the C code implements and
the expectations the GPU tests (tests/test_dist_gpu.py) check the C code against:

  * destination of a row = low32(fmix64(key)) mod world                    (oracle dest_of = mdb_dev_partition_by_dest)
  * rows with a NULL key stay home, or all go to the destination of key 0  (MDB_DIST_KEEP_NULL_KEYS)
  * ONE count exchange of 3 counters per peer: rows, bit mask of the columns that carry NULL bits, status - a rank that
    failed says so there and every rank leaves together
  * columns travel as opaque 8-byte cells in send order, one uneven all-to-all each; NULL bits of all columns as ONE extra
    8-byte word per row, only when some rank has a NULL bitmap
  * received order = (source rank, position in that rank's send order)

Nothing under midoridb_amd/ imports this file."""
import numpy as np
import torch
import torch.distributed as dist

from oracle import np_oracle as orc

KEEP_NULL_KEYS = 1


class ExchangeFailed(RuntimeError):
    pass


def _alltoallv(send, scnt, rcnt):
    recv = torch.empty(int(sum(rcnt)), dtype=torch.int64)
    dist.all_to_all_single(recv, torch.from_numpy(np.ascontiguousarray(send)), [int(c) for c in rcnt], [int(c) for c in scnt])
    return recv.numpy()


def shuffle_rows(keys, key_nulls, cols, flags=0, fail=False):
    """keys: int64[n]; key_nulls: bool[n] or None; cols: list of (8-byte values ndarray, nulls bool ndarray or None, rid or None).
    -> ([(values int64 view, nulls or None)], received rows).  fail=True: this rank reports a failure with the counts."""
    world = dist.get_world_size()
    n = len(keys)
    k = np.asarray(keys, dtype=np.int64)
    if flags & KEEP_NULL_KEYS and key_nulls is not None:
        k = np.where(key_nulls, 0, k)
        valid = np.arange(n)
    else:
        valid = np.arange(n) if key_nulls is None else np.nonzero(~np.asarray(key_nulls))[0]
    d = orc.dest_of(k[valid], world)
    order = np.argsort(d, kind="stable")
    pos = valid[order]					# send position -> stream position
    scnt = np.bincount(d, minlength=world)
    colmask = sum(1 << c for c, col in enumerate(cols) if col[1] is not None)
    if fail:
        scnt[:] = 0
        pos = pos[:0]
    send = np.empty(3 * world, dtype=np.int64)
    send[0::3], send[1::3], send[2::3] = scnt, colmask, int(fail)
    recv = torch.empty(3 * world, dtype=torch.int64)
    dist.all_to_all_single(recv, torch.from_numpy(send))
    recv = recv.numpy()
    rcnt, gmask, failed = recv[0::3], int(np.bitwise_or.reduce(recv[1::3])), recv[2::3]
    if failed.any():
        raise ExchangeFailed(f"rank {int(np.nonzero(failed)[0][0])} failed; nothing was exchanged")
    out = []
    cells = []
    for values, nulls, rid in cols:
        rows = pos if rid is None else np.asarray(rid)[pos]
        cells.append(_alltoallv(np.asarray(values).view(np.int64)[rows], scnt, rcnt))
    got_nulls = [None] * len(cols)
    if gmask:
        mask = np.zeros(len(pos), dtype=np.int64)
        for c, (values, nulls, rid) in enumerate(cols):
            if nulls is not None:
                rows = pos if rid is None else np.asarray(rid)[pos]
                mask |= np.asarray(nulls)[rows].astype(np.int64) << c
        rmask = _alltoallv(mask, scnt, rcnt)
        for c in range(len(cols)):
            if gmask >> c & 1:
                got_nulls[c] = (rmask >> c & 1).astype(bool)
    for c in range(len(cols)):
        out.append((cells[c], got_nulls[c]))
    return out, int(rcnt.sum())


def join_pairs(keys_l, null_l, cols_l, keys_r, null_r, cols_r):
    """mdb_dist_join_pairs: both tables shuffled by key, joined locally -> (key, left columns, right columns) of the joined rows"""
    lo, nl = shuffle_rows(keys_l, null_l, [(keys_l, None, None)] + list(cols_l))
    ro, nr = shuffle_rows(keys_r, null_r, [(keys_r, None, None)] + list(cols_r))
    pl, pr = orc.join_pairs(lo[0][0], None, ro[0][0], None)
    take = lambda side, idx: [(v[idx], None if nb is None else nb[idx]) for v, nb in side[1:]]  # noqa: E731
    return lo[0][0][pl], take(lo, pl), take(ro, pr)


def join_group_count(keys_l, null_l, keys_r, null_r):
    """mdb_dist_join_group_count: -> (group keys, counts, joined rows) of the groups whose key hashes to this rank"""
    lo, _ = shuffle_rows(keys_l, null_l, [(keys_l, None, None)])
    ro, _ = shuffle_rows(keys_r, null_r, [(keys_r, None, None)])
    k, c, _, j = orc.join_group_count(lo[0][0], None, ro[0][0], None)
    return k, c, j
