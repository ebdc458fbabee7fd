#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes.

Reads: the request-size classes TCC_EA0_RDREQ_{32B,64B,128B}_sum -> 32*a + 64*b + 128*c bytes.  On gfx950 practically every memory-side
read request of these kernels is a 128-byte one; the derived counter FETCH_SIZE tallies those at 64 bytes, i.e. it under-reports by 2x
for EVERY kernel of this path, 4-B/lane loads included (round 1's "calibration" on k_blur -- FETCH_SIZE 277 MB against 286 MB of input
-- was a coincidence of that factor 2 with k_blur's 1.94x over-fetch).  Writes: WRITE_SIZE (32 / 64-byte write requests; equals the known
store volume of k_blur and k_pyr_base).  Writes profiles/<tag>_pmc_traffic.csv.
"""
import csv
import re
import sqlite3
import sys
from collections import defaultdict

read_db, write_db, tag = sys.argv[1], sys.argv[2], sys.argv[3]
W = {"TCC_EA0_RDREQ_32B_sum": 32.0, "TCC_EA0_RDREQ_64B_sum": 64.0, "TCC_EA0_RDREQ_128B_sum": 128.0, "WRITE_SIZE": 1024.0}


def per_kernel(db):
    c = sqlite3.connect(db)
    acc = defaultdict(lambda: [0.0, set()])
    for name, cn, val, did in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        if cn not in W:
            continue
        k = re.sub(r"<.*>", "", name.split("(")[0].replace("void ", "").replace("orbx::", ""))  # template instances share a row
        acc[k][0] += val * W[cn]
        acc[k][1].add(did)
    return {k: (v[0] / len(v[1]), len(v[1])) for k, v in acc.items()}


r, w = per_kernel(read_db), per_kernel(write_db)
rows = []
for k in sorted(set(r) | set(w)):
    if not k.startswith("k_"):
        continue
    rk, n = r.get(k, (0.0, 0))
    wk, _ = w.get(k, (0.0, 0))
    rows.append([k, n, int(rk), int(wk), int(rk + wk)])
with open(f"profiles/{tag}_pmc_traffic.csv", "w", newline="") as fh:
    wr = csv.writer(fh)
    wr.writerow(["kernel", "dispatches", "read_bytes_per_launch(32/64/128B request classes)", "write_bytes_per_launch(WRITE_SIZE)", "traffic_bytes_per_launch"])
    wr.writerows(rows)
print(open(f"profiles/{tag}_pmc_traffic.csv").read())
