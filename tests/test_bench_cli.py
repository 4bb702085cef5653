"""bench.py's launcher contract that can be checked without a GPU: `--gpus N` on a box with fewer than N GPUs must not
crash or hang - one JSON line that says why nothing was measured, exit code 0 (the ranks are child processes started
before this process touches a GPU; on a box with N GPUs the same entry point relays rank 0's line)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_enough_devices_is_refused_cleanly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box can actually run --gpus 2")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] is None and line["n_gpus"] == 2 and "visible GPUs" in line["skipped"]
    assert line["metric"].startswith("joined rows/sec")


# ---- bench.py's N > 1 code on a one-GPU box (round 5): `--transport test` starts N ranks on GPU 0 and carries the blocks through host
# memory (DistCtx.over_host_group), so everything the first multi-GPU run will execute - the launcher, the process group, DistCtx set-up,
# wire_format / set_key_ranges from "catalog" statistics, the weak and the strong leg, per-rank phase JSON, the line relay - has run before.
import pytest  # noqa: E402

CORE_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
             "config", "roofline", "pipeline", "cpu_baseline"}


def _bench(argv, timeout=900, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=timeout, env=env)
    return p


def _one_line(p):
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_north_star_bench_at_world_n_through_the_test_transport(world):
    one = _one_line(_bench(["--rows", "400000", "--steps", "2", "--warmup", "1", "--no-secondary", "--no-cpu-baseline"]))
    line = _one_line(_bench(["--gpus", str(world), "--transport", "test", "--rows", "400000", "--steps", "2", "--warmup", "1", "--verify",
                             "--no-cpu-baseline", "--peer-timeout", "240"]))
    missing = (set(one) - {"exchange_preview"}) - set(line)
    assert not missing and CORE_KEYS <= set(line), missing
    assert line["n_gpus"] == world and line["scaling"] == "weak" and line["value"] > 0
    cfg = line["config"]
    assert cfg["rccl_ranks_seen"] == "test transport" and cfg["rows_per_table_total"] == 400000 * world
    assert cfg["joined_rows"] == 400000 * world          # variant D: every B row has its one partner
    assert line["verified_vs_oracle"] is True
    ex = line["exchange"]
    assert ex["plan"] and ex["plan"]["world"] == world and len(ex["per_rank"]) == world
    assert sorted(r["rank"] for r in ex["per_rank"]) == list(range(world)) and all("first_level_ms" in r["phases_ms"] for r in ex["per_rank"])
    assert ex["per_link"]["bytes_per_peer_per_call"] == ex["plan"]["bytes_per_peer"] and ex["per_link"]["peers"] == world - 1
    # the plan that ran is the plan the N = 1 line previews for this world (same rows per GPU, the N-fold key ranges)
    prev = one["exchange_preview"]["by_world"][str(world)]
    assert prev["bytes_per_peer_per_call"] == ex["plan"]["bytes_per_peer"], (prev, ex["plan"])
    strong = line["strong_scaling"]
    assert strong["rows_per_table_per_gpu"] == 400000 // world and strong["joined_rows"] == 400000 and strong["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("config,world", [(4, 2), (4, 8), (5, 2), (5, 8)])
def test_config_4_and_5_bench_at_world_n_through_the_test_transport(config, world):
    rows = 300000
    line = _one_line(_bench(["--config", str(config), "--gpus", str(world), "--transport", "test", "--rows", str(rows), "--steps", "2",
                             "--warmup", "1", "--peer-timeout", "240"]))
    assert CORE_KEYS <= set(line), CORE_KEYS - set(line)
    assert line["n_gpus"] == world and line["value"] > 0 and line["config"]["rccl_ranks_seen"] == "test transport"
    # unique keys drawn from one permutation per table: every key of A has its partner in B (and C)
    assert line["config"]["joined_rows"] == rows * world
    if config == 5:
        assert line["join_only_form"]["joined_rows"] == rows * world


@pytest.mark.gpu
@pytest.mark.parametrize("fault", ["exit:1", "hang:0"])
def test_a_rank_that_loses_a_peer_exits_non_zero(fault):
    """one rank dies (exit) or stops taking part (hang) after the warm-up: its peers must not wait for ever - every rank exits, the
    launcher returns non-zero, well inside the test's timeout (the hanging rank is ended by its own watchdog: an exit, never a re-exec)"""
    import time
    t0 = time.time()
    p = _bench(["--gpus", "2", "--transport", "test", "--rows", "200000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                "--peer-timeout", "20"], timeout=600, env_extra={"MDB_BENCH_FAULT": fault})
    assert p.returncode != 0, p.stdout[-1000:]
    assert time.time() - t0 < 400
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"value"' in ln and '"value": null' not in ln]
