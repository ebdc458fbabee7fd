#!/usr/bin/env python3
"""Per-kernel averages of SQ counters from one or more rocprofv3 --pmc result .db files -> CSV (profiles/<tag>_pmc_sq_counters.csv).
Derived columns: VALU issue time (SQ_INSTS_VALU x 4 cycles / 1024 SIMDs) in us at 2.4 GHz, VALU instructions per wave."""
import csv
import re
import sqlite3
import sys
from collections import defaultdict

tag, dbs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for db in dbs:
    c = sqlite3.connect(db)
    for name, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
        k = re.sub(r"<.*>", "", name.split("(")[0].replace("orbx::", "").replace("void ", ""))
        a = acc[k][cn]
        a[0] += val
        a[1] += 1
counters = sorted({cn for k in acc for cn in acc[k]})
rows = []
for k in sorted(acc):
    if not k.startswith("k_"):
        continue
    r = {cn: acc[k][cn][0] / acc[k][cn][1] for cn in acc[k]}
    row = [k] + [int(r.get(cn, 0)) for cn in counters]
    valu, waves = r.get("SQ_INSTS_VALU", 0), r.get("SQ_WAVES", 0)
    row += [round(valu * 4 / 1024 / 2400.0, 1), round(valu / waves, 1) if waves else 0]
    rows.append(row)
out = f"profiles/{tag}_pmc_sq_counters.csv"
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel (avg per launch)"] + counters + ["valu_issue_us@2.4GHz", "valu_per_wave"])
    w.writerows(rows)
print(open(out).read())
