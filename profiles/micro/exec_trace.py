"""HIP API calls of one query_execute() of the headline statement (tables and result on the device), after warm-up:
    rocprofv3 --hip-trace --output-format csv -d gpurun_out/etrace -- python3 profiles/micro/exec_trace.py run
    python3 profiles/micro/exec_trace.py show gpurun_out/etrace"""
import csv, glob, os, sys, time
if sys.argv[1] == "run":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from midoridb_amd.query import DB
    n = 100_000_000
    SQL = "SELECT id_a, COUNT(*) FROM A INNER JOIN B ON id_a = id_b GROUP BY id_a;"
    with DB() as db:
        db.execute("CREATE TABLE A (id_a INT);")
        db.execute("CREATE TABLE B (id_b INT);")
        db.generate("A", n, 42)
        db.generate("B", n, 43, [n // 16])
        db.results_on_device(True)
        for i in range(6):
            if i == 5:
                time.sleep(0.05)        # (a gap in the trace: the last call follows it)
            t0 = time.perf_counter()
            db.query_device(SQL, copy=False)
            print("call %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3))
else:
    rows = []
    for fn in glob.glob(os.path.join(sys.argv[2], "**", "*hip_api_trace.csv"), recursive=True):
        with open(fn) as f:
            rows += list(csv.DictReader(f))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    cut = 0
    for i in range(1, len(rows)):
        if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 30_000_000:
            cut = i
    t0 = int(rows[cut]["Start_Timestamp"])
    for r in rows[cut:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  {r['Function']}")
    print("calls", len(rows) - cut, "span %.1f us" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e3))
