"""Ablations of k_leaf_wide12 (MDB_DEBUG_W12 bit mask: 1 no left counting, 2 no right counting, 4 no emit at all, 8 no record stores):
kernel times of variant U through bench.py."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for m in (sys.argv[1:] or ["0", "8", "4", "5", "6", "7"]):
    env = dict(os.environ, MDB_DEBUG_W12=m, MDB_LIBRARY=os.path.join(ROOT, "midoridb_amd", "csrc", "build", "libabl.so"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--variant", "U", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-secondary"],
                       env=env, capture_output=True, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if not lines:
        print("mask", m, "no line:", (p.stderr or p.stdout)[-400:])
        continue
    s = json.loads(lines[-1])
    print("mask", m, round(s["ms_per_step"], 3), {k: round(v["ms_per_step"], 4) for k, v in s["kernels"].items() if "wide12" in k or "order" in k}, flush=True)
