"""Summarise a rocprofv3 rocpd sqlite database: per-kernel count / total / average duration.
Usage: python scripts/rocpd_stats.py <results.db> [top_n]   (writes a markdown table to stdout)"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   f"from kernels group by {name_col} order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
for n, c, s, a, mn, mx in rows[:top]:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    n = n if (len(n) < 90 or __import__("os").environ.get("KRS_STATS_FULL_NAMES")) else n[:87] + "..."
    print(f"| {n} | {c} | {s/1e6:.3f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*s/total:.1f} |")
print(f"\ntotal kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
