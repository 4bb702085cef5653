"""Every environment knob the library reads is named in INTEGRATION.md or in a public header: a knob that exists only in the
source is a behaviour a maintainer cannot find."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_environment_knob_is_documented():
    knobs = set()
    for f in glob.glob(os.path.join(ROOT, "midoridb_amd", "csrc", "*")):
        if f.endswith((".hip", ".c", ".h")):
            knobs |= set(re.findall(r'(?:getenv|mdb_knob)\("(M[A-Z0-9_]+)"\)', open(f).read()))
    assert len(knobs) > 20
    docs = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        docs += open(h).read()
    missing = sorted(k for k in knobs if k not in docs)
    assert not missing, missing


def test_the_environment_is_read_in_one_place():
    """round 6: mdb_knob() (mdb_dev_core.hip) is the library's one reader of its MDB_* knobs - kept per process, dropped by
    mdb_dev_reload_knobs(); the device layer calls getenv nowhere else"""
    n = 0
    for f in glob.glob(os.path.join(ROOT, "midoridb_amd", "csrc", "*.hip")):
        n += len(re.findall(r'\bgetenv\(', open(f).read()))
    assert n <= 2, n

