"""Effective shader clock per kernel from a rocprofv3 --pmc GRBM_GUI_ACTIVE pass (development aid):
   clock = counter / dispatch duration (GRBM_GUI_ACTIVE counts cycles while the graphics pipe is busy; one
   value per dispatch).  Usage: python scripts/rocpd_clock.py <results.db> [substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else "krs::"
rows = db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration), avg(value * 1.0 / duration) "
                  "from counters_collection group by kernel_name, counter_name order by 1").fetchall()
for k, c, n, v, d, r in rows:
    if sub in k:
        k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        print(f"{k:70s} {c:18s} n={n:3d} value={v:14.1f} dur={d/1e3:9.1f} us  value/ns={r:7.3f}")
