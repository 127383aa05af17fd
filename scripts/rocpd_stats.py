#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (…_results.db) into the per-kernel table `rocprofv3 --stats` prints:
name, calls, total ns, average ns, percentage.   usage: rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys

db = sys.argv[1]
con = sqlite3.connect(db)
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in kcols else ("display_name" if "display_name" in kcols else "name")
q = f"""select s.{name_col}, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.{name_col} order by 3 desc"""
rows = cur.execute(q).fetchall()
total = sum(r[2] for r in rows) or 1
lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
for n, c, t, a, mn, mx in rows:
    lines.append(f'"{n}",{c},{t},{a:.1f},{mn},{mx},{100.0 * t / total:.2f}')
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
print(out)
