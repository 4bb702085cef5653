"""world_size-2 test of the multi-GPU exchange PROTOCOL on CPU (gloo), SURVEY.md 8e: the product's exchange
(midoridb_amd/csrc/mdb_dist.hip) runs HIP kernels and is tested with two ranks on the GPU box (tests/test_dist_gpu.py); here
its protocol - destinations, counts + NULL mask + status agreement, column transfers, received order - is modelled with the
oracle's operators (tests/_exchange_model.py), which also pins the expectations the GPU tests use."""
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
    assert "gloo shuffle rows ok" in r.stdout			# keys + payload + NULL bits by destination, failure agreed with the counts
    assert "gloo distributed payload join ok" in r.stdout		# 3-way join with DOUBLE/INT payload + GROUP BY (config 5 shape)
