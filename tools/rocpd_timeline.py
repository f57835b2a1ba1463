#!/usr/bin/env python3
"""Timeline of the last verification in a rocprofv3 rocpd database: every kernel from the last k_setup on,
start and end in microseconds relative to that k_setup's start.
usage: tools/rocpd_timeline.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = db.execute(f"select {name}, start, end from kernels order by start").fetchall()
last = max(i for i, r in enumerate(rows) if "k_setup" in r[0])
t0 = rows[last][1]
print("start_us,end_us,dur_us,kernel")
for n, a, b in rows[last:]:
    print("%.1f,%.1f,%.1f,%s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, n.split("(")[0].replace("xhe::", "")))
