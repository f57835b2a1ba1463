#!/usr/bin/env python3
"""Per-kernel averages of the counters of a rocprofv3 --pmc run (rocpd database), as CSV on stdout.
FETCH_SIZE / WRITE_SIZE are in KiB per dispatch as rocprofv3 reports them (no correction applied).
usage: tools/rocpd_pmc.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
                  "group by kernel_name, counter_name order by 4 desc").fetchall()
print("kernel,counter,dispatches,avg,min,max")
for r in rows:
    print('"%s",%s,%d,%.1f,%.1f,%.1f' % (r[0].split("(")[0], r[1], r[2], r[3], r[4], r[5]))
