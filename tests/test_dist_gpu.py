"""The multi-GPU exchange behind the C-ABI (include/mdb_dist.h, csrc/mdb_dist.hip) on the one-GPU test box:
world size 1 over RCCL, and world size 2 on the same GPU with a test transport (gloo through host memory) plugged into
struct mdb_dist_transport - the same C code path for counts, displacements, receive buffers, events and the local join."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, nproc, port, worker="_dist_gpu_worker.py", timeout=1200):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", worker), mode]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-4000:]
    return r.stdout


@pytest.mark.parametrize("world", [4, 8])
def test_every_sharded_operator_at_world_4_and_8_on_one_gpu(world):
    """N contexts on one device, bytes moved by the test transport: join + GROUP BY (one level with 2-byte words, the receiver's own
    level, 4096-digit senders, a 2^30-value window = BASELINE configs[3]'s plan), three and four tables in one exchange (configs[4]),
    GROUP BY of one column, the keys-only and the materialising join, row shuffles, skew -> fallback on every rank; each case
    asserts which branch of mdb_shard_plan_make ran (mdb_dist_last_plan)"""
    assert f"plans shapes world {world} ok" in _run("shapes", world, 29620 + world, "_dist_plans_worker.py", 900)


@pytest.mark.parametrize("world", [4, 8])
def test_query_execute_sharded_at_world_4_and_8_on_one_gpu(world):
    """the statement shapes of the world-2 test and BASELINE configs[4]'s own statement (three tables, DOUBLE payload, one key,
    GROUP BY; and its join-only form) through query_execute() in sharded mode"""
    assert f"plans sql world {world} ok" in _run("sql", world, 29640 + world, "_dist_plans_worker.py", 900)


@pytest.mark.parametrize("world", [2, 4])
def test_a_rank_whose_first_level_fails_takes_every_rank_out_with_an_error(world):
    """fault injection (MDB_DIST_FAULT): the failing rank posts what its peers wait for and says so in its region counters; an error
    on every rank, nobody hangs, the next call works; ranks in different collective calls find out before anything is posted"""
    assert f"plans fault world {world} ok" in _run("fault", world, 29660 + world, "_dist_plans_worker.py", 600)


def test_sharded_join_group_count_over_rccl():
    import torch
    n = max(1, torch.cuda.device_count())
    assert f"dist rccl world {n} ok" in _run("rccl", n, 29611)


def test_sharded_join_group_count_two_ranks_one_gpu_test_transport():
    assert "dist gloo world 2 ok" in _run("gloo", 2, 29612)


def test_query_execute_sharded_any_equi_join_two_ranks_one_gpu():
    """every equi-join shape through query_execute() at world size 2 (rows exchanged by mdb_dist_shuffle_rows) vs oracle/naive.py"""
    assert "sharded sql world 2 ok" in _run("sql", 2, 29613)


def test_query_execute_in_sharded_mode_world1(tmp_path):
    """query_execute() with MIDORIDB_WORLD_SIZE set runs the fused plan through mdb_dist_join_group_count_alloc (RCCL
    communicators created inside database code from the id file; world size 1 on the test box): same groups and counts as
    the single-GPU plan, chained over a third table, SELECT COUNT(*) all-reduced, materialising joins exchanged, joins without an
    equi-join key answered by replicating the new table."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    script = r'''
import numpy as np, sys
sys.path.insert(0, %r)
from midoridb_amd.query import DB, QueryError
from oracle import np_oracle as orc
rng = np.random.default_rng(4)
n = 300_000
a, b, c = rng.integers(0, 50_000, n), rng.integers(0, 60_000, n + 7), rng.integers(0, 50_000, 1000)
with DB() as db:
    for t, col, v in (("A", "id_a", a), ("B", "id_b", b), ("C", "id_c", c)):
        db.execute(f"CREATE TABLE {t} ({col} INT);")
        db.append_columns(t, [v.astype(np.int64)])
    r = db.query("SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a;")
    ek, ec, _, ej = orc.join_group_count(a, None, b, None)
    got = dict(zip(r.columns[r.names.index("A.id_a")].tolist(), r.columns[r.names.index("COUNT(*)")].tolist()))
    assert got == dict(zip(ek.tolist(), ec.tolist())) and r.joined_rows == ej
    assert db.query("SELECT COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b;").rows() == [(ej,)]
    r3 = db.query("SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b INNER JOIN C ON A.id_a = C.id_c GROUP BY id_a;")
    cc = dict(zip(*[x.tolist() for x in np.unique(c, return_counts=True)]))
    exp3 = {k: v * cc[k] for k, v in zip(ek.tolist(), ec.tolist()) if k in cc}
    assert dict(zip(r3.columns[r3.names.index("A.id_a")].tolist(), r3.columns[r3.names.index("COUNT(*)")].tolist())) == exp3
    assert db.query("SELECT COUNT(*) FROM A WHERE id_a < 10;").rows() == [(int((a < 10).sum()),)]
    r = db.query("SELECT * FROM A INNER JOIN C ON A.id_a = C.id_c;")		# a materialising join: exchanged over RCCL like any other
    pl, pr = orc.join_pairs(a, None, c, None)
    assert sorted(r.columns[r.names.index("A.id_a")].tolist()) == sorted(a[pl].tolist()) and r.nrows == len(pl)
    few = db.query("SELECT id_a, id_c FROM A, C WHERE id_a < 3 AND id_c < 5;")		# no equi-join key: C is replicated (mdb_dist_broadcast_rows)
    assert few.nrows == int((a < 3).sum()) * int((c < 5).sum())
    try:
        db.query("SELECT * FROM A, C;")
        raise SystemExit("3 x 10^8 pairs of a cross join must be refused")
    except QueryError as e:
        assert "too large" in str(e)
print("sharded query_execute ok")
''' % ROOT
    env = dict(os.environ, MIDORIDB_WORLD_SIZE="1", MIDORIDB_RANK="0", MIDORIDB_DIST_ID_FILE=str(tmp_path / "mdb_id"),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", script], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "sharded query_execute ok" in r.stdout, r.stdout[-4000:]
