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
