#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel table: calls, total/avg/min/max ns, %.
usage: tools/rocpd_summary.py results.db > profiles/xxx_kernel_stats.csv"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
tot = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for r in rows:
    print('"%s",%d,%d,%.1f,%d,%d,%.2f' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
