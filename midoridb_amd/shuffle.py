"""Exchange helpers over torch.distributed for the rows of the path that carry PAYLOAD columns (BASELINE configs 4 / 5:
joins that materialise rows): hash-partition by destination GPU with source row ids -> payload gathered into send order ->
one uneven all-to-all per column (RCCL; backend "nccl" IS RCCL on ROCm) -> local joins (TableShuffle).

The north-star operator itself (join + GROUP BY key + COUNT(*), keys only) is NOT here any more: its exchange lives
behind the C-ABI (include/mdb_dist.h, csrc/mdb_dist.hip - RCCL communicators, counts, all-to-all and the split local join in
C), bound by midoridb_amd/dist.py and used by bench.py and by query_execute()'s sharded mode.  DistributedJoinGroupCount below
is the backend-neutral restatement of that sequence which the world_size-2 gloo test on CPU runs with the oracle's
partition / join functions plugged in.
"""
import torch
import torch.distributed as dist


class KeyExchange:
    """counts all-to-all (nGPU int64) followed by one uneven all_to_all_single of the keys.  The tiny count
    exchanges run on their own process group: on the data communicator they would queue behind a key transfer
    that is still in flight and stall the host (they are read back with .tolist())."""

    def __init__(self, world, device, count_group=None):
        self.world = world
        self.device = device
        self.count_group = count_group

    def exchange(self, send_keys, send_counts, recv_buf=None, async_op=False, recv_offset=0):
        """send_keys: keys grouped by destination rank; send_counts: list[int] of len world.
        Returns (recv_keys tensor, recv_counts list, work handle or None); the received keys land at
        recv_buf[recv_offset:]."""
        cin = torch.tensor(send_counts, dtype=torch.int64, device=self.device)
        cout = torch.empty(self.world, dtype=torch.int64, device=self.device)
        dist.all_to_all_single(cout, cin, group=self.count_group)
        recv_counts = [int(x) for x in cout.tolist()]
        total = sum(recv_counts)
        if recv_buf is None or recv_buf.numel() < recv_offset + total:
            if recv_buf is not None and recv_offset:
                raise RuntimeError("receive buffer too small for a chunked exchange")
            recv_buf = torch.empty(max(total, 1), dtype=send_keys.dtype, device=self.device)
        recv = recv_buf[recv_offset:recv_offset + total]
        work = dist.all_to_all_single(recv, send_keys[:sum(send_counts)], recv_counts, list(send_counts), async_op=async_op)
        return recv, recv_counts, work


class TableShuffle:
    """Moves one table to its owners: rows travel to rank hash(key) mod world with their payload columns
    (BASELINE configs 4 and 5: joins that materialise rows, DOUBLE payload included - payload bytes are moved as
    opaque 8-byte cells, so they stay bit-exact).  partition_fn(keys) -> (keys grouped by destination, counts,
    source row of every entry); gather_fn(column, rows) -> column in send order.  One counts exchange, then one
    uneven all_to_all_single per column, all asynchronous and waited for together."""

    def __init__(self, world, device, partition_fn, gather_fn):
        self.world = world
        self.device = device
        self.ex = KeyExchange(world, device)
        self.partition_fn = partition_fn
        self.gather_fn = gather_fn

    def run(self, keys, payload=(), with_origin=False, rank=0):
        """-> (received keys, [received payload columns], received origin ids or None).  origin = (source rank << 32)
        | source row: the global identity of a row after the exchange."""
        send_keys, counts, rows = self.partition_fn(keys)
        recv_keys, recv_counts, work = self.ex.exchange(send_keys, counts, async_op=True)
        works = [work]
        total, sent = sum(recv_counts), sum(counts)
        cols = [self.gather_fn(c, rows) for c in payload]
        if with_origin:
            cols.append(rows.to(torch.int64) + (int(rank) << 32))
        outs = []
        for c in cols:
            r = torch.empty(max(total, 1), dtype=c.dtype, device=self.device)[:total]
            works.append(dist.all_to_all_single(r, c[:sent], recv_counts, list(counts), async_op=True))
            outs.append(r)
        for w in works:
            if w is not None:
                w.wait()
        origin = outs.pop() if with_origin else None
        return recv_keys, outs, origin


class DistributedJoinGroupCount:
    """One rank's share of  A JOIN B ON id_a = id_b GROUP BY id_a COUNT(*)  over `world` GPUs.

    Table A's all-to-all travels over xGMI while table B is partitioned by destination, and table B's while the
    received A is hashed and radix-partitioned locally - per-link xGMI bandwidth, not HBM, is what bounds the
    exchange (0.7 of every table leaves each GPU at 8 ranks).  `chunks` > 1 additionally splits every table into
    pieces whose transfers overlap the partitioning of the next piece; measured on one GPU every extra round costs
    ~0.7 ms of host synchronisation (counts must reach the host before a transfer can be posted), more than the
    overlap can win at the transfer sizes of this workload, so the default is one piece."""

    def __init__(self, dev, world, rank, rows_per_rank, partition_fn=None, join_fn=None, device=None, chunks=None, wire32=False,
                 widen_fn=None):
        """wire32: both key columns are known (column statistics, mdb_dev_key_range on every rank) to fit 32 bits: the
        keys then cross xGMI as 4-byte integers - half the bytes of the step that bounds multi-GPU throughput - and
        are widened on arrival (one streaming kernel per table, overlapped with the other table's transfer)."""
        self.dev = dev
        self.world = world
        self.rank = rank
        device = device if device is not None else dev.device
        self.chunks = max(1, int(chunks if chunks is not None else 1))
        count_group = dist.new_group() if self.chunks > 1 or world > 1 else None
        self.ex = KeyExchange(world, device, count_group)
        self.wire32 = bool(wire32)
        cap = int(rows_per_rank * 1.3) + 4096
        self.send_a = torch.empty(max(rows_per_rank, 2) + 2, dtype=torch.int64, device=device)
        self.send_b = torch.empty(max(rows_per_rank, 2) + 2, dtype=torch.int64, device=device)
        self.recv_a = torch.empty(cap, dtype=torch.int64, device=device)
        self.recv_b = torch.empty(cap, dtype=torch.int64, device=device)
        if self.wire32:
            self.recv_a32 = torch.empty(cap, dtype=torch.int32, device=device)
            self.recv_b32 = torch.empty(cap, dtype=torch.int32, device=device)
        self.partition_fn = partition_fn or (lambda keys, out: dev.partition_by_dest(keys, None, world, out=out, keys32=self.wire32))
        self.widen_fn = widen_fn or (lambda src32, out: dev.widen32(src32, out=out))
        self.join_fn = join_fn      # None: split device operator (begin on A while B is still in flight)
        self.n_r_max = cap

    def _send_table(self, keys, send_buf, recv_buf):
        """partition by destination + exchange, piece by piece -> (received keys (view of recv_buf), work handles)"""
        n = keys.numel()
        per = -(-n // self.chunks)
        per += per & 1                  # even piece starts keep the 16-byte alignment the partition kernel loads with
        if self.wire32:
            send_buf = send_buf.view(torch.int32)
        works, off = [], 0
        for c in range(self.chunks):
            lo, hi = min(n, c * per), min(n, (c + 1) * per)
            # every rank runs every round, also with an empty piece: the collectives must pair up across ranks
            s, cnt = self.partition_fn(keys[lo:hi], send_buf[lo:hi] if hi > lo else send_buf[:1])
            _, rc, w = self.ex.exchange(s, cnt, recv_buf, async_op=True, recv_offset=off)
            off += sum(rc)
            works.append(w)
        return recv_buf[:off], works

    @staticmethod
    def _wait(works):
        for w in works:
            if w is not None:
                w.wait()

    def run(self, a, b, out=None):
        # table A's pieces travel over xGMI while the next piece / table B is being partitioned
        ra, wa = self._send_table(a, self.send_a, self.recv_a32 if self.wire32 else self.recv_a)
        rb, wb = self._send_table(b, self.send_b, self.recv_b32 if self.wire32 else self.recv_b)
        self._wait(wa)
        if self.join_fn is not None:
            # injected join (the gloo tests' oracle): works on int64 columns
            if self.wire32:
                ra = self.widen_fn(ra, self.recv_a)
            self._wait(wb)
            if self.wire32:
                rb = self.widen_fn(rb, self.recv_b)
            k, c, f, j = self.join_fn(ra, rb, out)
        else:
            # A has arrived: hash + partition it locally while B's all-to-all is still running.  Keys that crossed in
            # the 4-byte wire format are consumed as they are (int32 entry points of the operator): no widening pass
            self.dev.join_group_count_begin(ra, None, self.n_r_max)
            self._wait(wb)
            k, c, f, j = self.dev.join_group_count_finish(rb, None, out=out)
        self.last = (k, c, f)
        return k.numel() if hasattr(k, "numel") else len(k), j


def algorithmic_bytes(kernel, n, groups, narrow=False):
    """Algorithmic HBM bytes of ONE launch of `kernel` on tables of n rows per GPU producing `groups`
    result groups: what the launch must read once and write once, independent of how it is
    implemented.  A partition kernel name covers the launch for table A (carries 4-byte row ids) and the
    one for table B (keys only); the figure is their average.  narrow: the form for int32-range keys (the left
    side's hash and row id share one 8-byte word, the right side is a 4-byte hash after level 0) - fewer bytes
    have to move, and the figure says so rather than crediting the kernels with the wide form's bytes."""
    key, rid, g = 8 * n, 4 * n, groups
    if narrow:
        h32 = 4 * n
        table = {
            "part_scatter_l0": ((key + key) + (key + h32)) / 2,            # read key; write hash|row id (A), 4-byte hash (B)
            "part_scatter_l1": ((2 * key) + (2 * h32)) / 2,
            "leaf_join_group_count": key + h32 + 8 * g,
            "leaf_group_count": key + 8 * g,
        }
        if kernel in table:
            return float(table[kernel])
    table = {
        "part_hist_l0": key,                                            # read the raw keys
        "part_scatter_l0": ((key + key + rid) + (key + key)) / 2,      # read key, write hash (+ row id)
        "part_scatter_l1": ((2 * (key + rid)) + (2 * key)) / 2,        # move hash (+ row id) into its leaf
        "leaf_join_group_count": (key + rid) + key + 8 * g,            # read both partitioned tables, write one record per group
        "leaf_group_count": (key + rid) + 8 * g,
        "sort_hist_l0": 8 * g,                                          # ordering sort over the group records
        "sort_scatter_l0": 16 * g,
        "sort_scatter_l1": 16 * g,
        "order_leaf": 8 * g + 12 * g,                                   # read records, write (first row id, COUNT)
        "gather64": 4 * g + 8 * g + 8 * g,                              # row id -> key of every group
    }
    return float(table.get(kernel, key))
