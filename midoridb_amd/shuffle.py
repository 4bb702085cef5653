"""Multi-GPU form of the north-star pipeline: hash-partition by destination GPU -> one all-to-all
per table over xGMI (RCCL through torch.distributed; backend "nccl" IS RCCL on ROCm) -> local
LDS hash join + group count.  No other collective is on the data path: groups are disjoint
across ranks because both tables are partitioned by the join key.

The exchange logic is backend-neutral so the world_size-2 gloo tests on CPU run exactly this
code with the oracle's partition/join functions plugged in instead of the device operators.
"""
import torch
import torch.distributed as dist


class KeyExchange:
    """counts all-to-all (nGPU int64) followed by one uneven all_to_all_single of the keys."""

    def __init__(self, world, device):
        self.world = world
        self.device = device

    def exchange(self, send_keys, send_counts, recv_buf=None, async_op=False):
        """send_keys: keys grouped by destination rank; send_counts: list[int] of len world.
        Returns (recv_keys tensor, recv_counts list, work handle or None)."""
        cin = torch.tensor(send_counts, dtype=torch.int64, device=self.device)
        cout = torch.empty(self.world, dtype=torch.int64, device=self.device)
        dist.all_to_all_single(cout, cin)
        recv_counts = [int(x) for x in cout.tolist()]
        total = sum(recv_counts)
        if recv_buf is None or recv_buf.numel() < total:
            recv_buf = torch.empty(max(total, 1), dtype=send_keys.dtype, device=self.device)
        recv = recv_buf[:total]
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
    """One rank's share of  A JOIN B ON id_a = id_b GROUP BY id_a COUNT(*)  over `world` GPUs."""

    def __init__(self, dev, world, rank, rows_per_rank, partition_fn=None, join_fn=None, device=None):
        self.dev = dev
        self.world = world
        self.rank = rank
        device = device if device is not None else dev.device
        self.ex = KeyExchange(world, device)
        cap = int(rows_per_rank * 1.3) + 4096
        self.send_a = torch.empty(max(rows_per_rank, 1), dtype=torch.int64, device=device)
        self.send_b = torch.empty(max(rows_per_rank, 1), dtype=torch.int64, device=device)
        self.recv_a = torch.empty(cap, dtype=torch.int64, device=device)
        self.recv_b = torch.empty(cap, dtype=torch.int64, device=device)
        self.partition_fn = partition_fn or (lambda keys, out: dev.partition_by_dest(keys, None, world, out=out))
        self.join_fn = join_fn      # None: split device operator (begin on A while B is still in flight)
        self.n_r_max = cap

    def run(self, a, b, out=None):
        # table A's keys travel over xGMI while table B is being partitioned
        sa, ca = self.partition_fn(a, self.send_a)
        ra, _, wa = self.ex.exchange(sa, ca, self.recv_a, async_op=True)
        sb, cb = self.partition_fn(b, self.send_b)
        rb, _, wb = self.ex.exchange(sb, cb, self.recv_b, async_op=True)
        if self.join_fn is not None:
            for w in (wa, wb):
                if w is not None:
                    w.wait()
            k, c, f, j = self.join_fn(ra, rb, out)
        else:
            # A has arrived: hash + partition it locally while B's all-to-all is still running
            if wa is not None:
                wa.wait()
            self.dev.join_group_count_begin(ra, None, self.n_r_max)
            if wb is not None:
                wb.wait()
            k, c, f, j = self.dev.join_group_count_finish(rb, None, out=out)
        self.last = (k, c, f)
        return k.numel() if hasattr(k, "numel") else len(k), j


def algorithmic_bytes(kernel, n, groups):
    """Algorithmic HBM bytes of ONE launch of `kernel` on tables of n rows per GPU producing `groups`
    result groups: what the launch must read once and write once, independent of how it is
    implemented.  A partition kernel name covers the launch for table A (carries 4-byte row ids) and the
    one for table B (keys only); the figure is their average."""
    key, rid, g = 8 * n, 4 * n, groups
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
