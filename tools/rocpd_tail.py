#!/usr/bin/env python3
"""The kernels of the last <ms> milliseconds of a rocprofv3 rocpd database (default 35): start, end, duration in microseconds
relative to the first of them, the queue and stream ids when the database has them, the kernel's name.
usage: tools/rocpd_tail.py <results.db> [ms]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
span = float(sys.argv[2]) if len(sys.argv) > 2 else 35.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
extra = [c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols]
rows = db.execute("select %s, start, end%s from kernels order by start" % (name, "".join(", " + c for c in extra))).fetchall()
t_end = max(r[2] for r in rows)
rows = [r for r in rows if r[1] >= t_end - span * 1e6]
t0 = rows[0][1]
print("start_us,end_us,dur_us,%skernel" % "".join(c + "," for c in extra))
for r in rows:
    print("%.1f,%.1f,%.1f,%s%s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, "".join(str(x) + "," for x in r[3:]), r[0].split("(")[0].replace("xhe::", "")))
