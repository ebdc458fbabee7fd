#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch).

Units/corrections as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: both counters are KiB (x1024 -> bytes);
FETCH_SIZE under-reports by 2x ONLY for wide (16 B/lane) coalesced streams and is "uncalibrated for other widths:
calibrate on a known byte count in your own access pattern".  Calibration for this code base: k_blur streams its
input exactly once with 4 B/lane loads (known: P = 1,117,367 px/frame x 128 frames = 143.0 MB) and reports
FETCH_SIZE = 135,301 KiB = 138.5 MB, i.e. factor 0.97 -> our 4-B/lane kernels need NO doubling.  traffic =
(FETCH_SIZE + WRITE_SIZE) x 1024; the doubled-fetch figure is kept in the CSV as an upper bound.
Writes profiles/<tag>_pmc_traffic.csv (bench.py --pmc measures the dominant kernel's traffic itself, in its own run).
"""
import csv
import re
import json
import sqlite3
import sys
from collections import defaultdict

fetch_db, write_db, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    acc = defaultdict(lambda: [0.0, 0])
    for name, val in c.execute("select kernel_name, value from counters_collection where counter_name=?", (counter,)):
        k = re.sub(r"<.*>", "", name.split("(")[0].replace("void ", "").replace("orbx::", ""))  # template instances share a row
        acc[k][0] += val
        acc[k][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


f = per_kernel(fetch_db, "FETCH_SIZE")
w = per_kernel(write_db, "WRITE_SIZE")
rows, js = [], {}
for k in sorted(set(f) | set(w)):
    if not k.startswith("k_"):
        continue
    fk, n = f.get(k, (0.0, 0))
    wk, _ = w.get(k, (0.0, 0))
    raw = (fk + wk) * 1024
    corr = (2 * fk + wk) * 1024
    rows.append([k, n, round(fk, 1), round(wk, 1), int(raw), int(corr)])
    js[k] = int(raw)
with open(f"profiles/{tag}_pmc_traffic.csv", "w", newline="") as fh:
    wr = csv.writer(fh)
    wr.writerow(["kernel", "dispatches", "FETCH_SIZE_KiB_avg", "WRITE_SIZE_KiB_avg", "traffic_bytes_per_launch(fetch+write)", "upper_bound(2*fetch+write)"])
    wr.writerows(rows)
print(open(f"profiles/{tag}_pmc_traffic.csv").read())
