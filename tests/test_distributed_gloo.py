"""world_size-2 test of the multi-GPU exchange logic on CPU (gloo), SURVEY.md 8e."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hash_partition_all_to_all_join_world2():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tests", "_gloo_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "gloo distributed join ok" in r.stdout
    assert "gloo distributed payload join ok" in r.stdout		# 3-way join with DOUBLE/INT payload + GROUP BY (config 5 shape)
