#!/usr/bin/env python3
"""Kernel statistics from a rocprofv3 rocpd database (the --kernel-trace --stats output of ROCm 7):
calls, total / average / min / max duration per kernel, as CSV on stdout.
usage: tools/rocpd_stats.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
dur = "duration" if "duration" in cols else "(end - start)"
rows = db.execute(f"select {name}, count(*), sum({dur}), avg({dur}), min({dur}), max({dur}) from kernels group by {name} order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent")
for r in rows:
    print('"%s",%d,%d,%.0f,%d,%d,%.2f' % (r[0].split("(")[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total))
