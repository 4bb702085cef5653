"""Plans as data (round 6): tests/golden/plans.json holds what the operators' decision code answers for the statistics of every BASELINE
configuration's tables - with catalog statistics and as a raw caller's first call, at world 1 / 2 / 8 for the sharded forms.  No GPU:
mdb_dev_explain_* run the operators' own entry points on a context without a device and stop in front of the first launch;
mdb_dist_plan_preview is a host computation.  A heuristic that changes shows HERE, as a diff (python tests/golden/make_plans.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def test_plans_are_what_the_committed_file_says():
    import make_plans
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "plans.json")))
    got = json.loads(json.dumps(make_plans.build()))
    assert set(got["single_gpu"]) == set(want["single_gpu"]) and set(got["sharded"]) == set(want["sharded"])
    for section in ("single_gpu", "sharded"):
        for name in want[section]:
            assert got[section][name] == want[section][name], (section, name, got[section][name], want[section][name])


def test_the_plans_of_the_baseline_configurations():
    """what the file must say, whatever else changes: the headline prunes and writes its records into the ordering ranges on a first
    statement; two primary keys leave as bits without a pilot; nothing samples when the catalog hands its statistics over"""
    p = json.load(open(os.path.join(ROOT, "tests", "golden", "plans.json")))["single_gpu"]
    d = p["configs[2] variant D (B 16x duplicated in the lowest sixteenth of A's range)"]
    assert d["with_catalog_statistics"]["minmax_pruned"] == 1 and d["with_catalog_statistics"]["ranged_order"] == 1 and d["with_catalog_statistics"]["levels"] == 1
    assert d["raw_caller_first_call"]["ranged_order"] == 0 and d["raw_caller_first_call"]["samples"] == 1
    u = p["configs[2] variant U (unique keys both sides)"]
    assert u["with_catalog_statistics"]["digits"] == 4096 and u["with_catalog_statistics"]["groups_as_bits"] == 3
    assert u["raw_caller_first_call"]["groups_as_bits"] == 1      # (a pilot launch decides)
    for name, e in p.items():
        if "with_catalog_statistics" in e:
            assert e["with_catalog_statistics"]["samples"] == 0 and e["with_catalog_statistics"]["retries"] == 0, name
    assert p["GROUP BY over a primary key, 10^8 rows"]["with_catalog_statistics"]["group_form"] == 3
    assert p["configs[4] join-only form, table by table: payload join, 10^8 rows"]["with_catalog_statistics"]["payload_form"] == 3
    m = p["configs[4] join-only form: B and C in one payload join, 10^8 rows"]["with_catalog_statistics"]
    assert m["payload_form"] == 3 and m["payload_tables"] == 2 and m["digits"] == 8192
    assert p["configs[1] join + payload, 10^7 rows, two cells"]["with_catalog_statistics"]["payload_form"] == 1
    assert p["README query, 6 rows per table"]["with_catalog_statistics"]["small_form"] == 1
