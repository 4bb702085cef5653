"""Sanitizer run of the host C (midoridb_amd/csrc/mdb_*.c: SQL front end, plan builder, catalog + storage, executor glue,
result cursor) and of the oracle's C restatement: `make asan` builds both with AddressSanitizer + UndefinedBehaviorSanitizer
(the reference's debug build uses -fsanitize=bounds + FORTIFY, reference scripts/config.mk:33-46) and the CPU test files
that drive them run in a child interpreter with the sanitizer runtime preloaded.  CPU only - never on the GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_suite_under_address_and_undefined_behaviour_sanitizers():
    if not shutil.which("gcc") or not shutil.which("make") or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("needs gcc, make and hipcc")
    import torch
    if torch.cuda.is_available():
        pytest.skip("sanitizer runs are for the CPU build only")
    for d, target in ((os.path.join(ROOT, "midoridb_amd", "csrc"), "asan"), (os.path.join(ROOT, "oracle"), "asan")):
        r = subprocess.run(["make", "-j", "8", target], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-3000:]
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=libasan, MDB_LIBRARY=os.path.join(ROOT, "midoridb_amd", "libmidoridb_amd_asan.so"),
               MDB_ORACLE_LIBRARY=os.path.join(ROOT, "oracle", "liboracle_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_cpu_frontend_abi.py"), os.path.join(ROOT, "tests", "test_oracle_pinning.py")],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "AddressSanitizer" not in r.stdout and "runtime error" not in r.stdout, r.stdout[-4000:]
