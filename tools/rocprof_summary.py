#!/usr/bin/env python3
"""Dump the per-kernel statistics of a rocprofv3 results .db (view top_kernels) as CSV for profiles/."""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, total, avg, pct in cur:
        w.writerow([name.split("(")[0], calls, round(total, 3), round(avg, 3), round(pct, 2)])
print(open(out).read())
