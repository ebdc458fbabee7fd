#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch).

Corrections as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE is doubled (it tallies 128-B
requests at 64 B for wide coalesced streams -- narrower access patterns are uncalibrated, so the doubled figure is an
upper estimate), WRITE_SIZE is taken as is; both x1024 -> bytes.  Writes profiles/<tag>_pmc_traffic.csv and
profiles/pmc_traffic.json (average corrected bytes per launch, read by bench.py for roofline.traffic).
"""
import csv
import json
import sqlite3
import sys
from collections import defaultdict

fetch_db, write_db, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    acc = defaultdict(lambda: [0.0, 0])
    for name, val in c.execute("select kernel_name, value from counters_collection where counter_name=?", (counter,)):
        k = name.split("(")[0].replace("orbx::", "")
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
    js[k] = int(corr)
with open(f"profiles/{tag}_pmc_traffic.csv", "w", newline="") as fh:
    wr = csv.writer(fh)
    wr.writerow(["kernel", "dispatches", "FETCH_SIZE_KiB_avg", "WRITE_SIZE_KiB_avg", "raw_bytes_per_launch", "corrected_bytes_per_launch(2*fetch+write)"])
    wr.writerows(rows)
json.dump(js, open("profiles/pmc_traffic.json", "w"), indent=1)
print(open(f"profiles/{tag}_pmc_traffic.csv").read())
