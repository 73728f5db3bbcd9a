"""Per-kernel PMC counter averages from a rocprofv3 rocpd sqlite database (development aid)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
views = [n for n in names if "pmc" in n.lower() or "counter" in n.lower()]
if len(sys.argv) > 2 and sys.argv[2] == "schema":
    for v in views:
        print(v, [r[1] for r in cur.execute(f"pragma table_info({v})")])
    sys.exit(0)
view = "counters_collection" if "counters_collection" in names else views[0]
cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "name" in c and "counter" not in c][0]
ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
vcol = "value" if "value" in cols else [c for c in cols if "value" in c][0]
for k, c, n, avg in cur.execute(f"select {kcol}, {ccol}, count(*), avg({vcol}) from {view} group by {kcol}, {ccol} order by 4 desc"):
    if "krs::" in k:
        print(k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80], c, n, round(avg, 1))
