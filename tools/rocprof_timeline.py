#!/usr/bin/env python3
"""Timeline of one steady-state step from a rocprofv3 --kernel-trace .db: kernels sorted by start time with queue id,
start offset (us), duration (us) and the idle gap to the previous kernel on the same queue."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(c.execute(f"select name, start, end, {qcol or '0'} from kernels order by start"))
# last occurrence of k_pyr_base but one = start of a steady-state step
bases = [i for i, r in enumerate(rows) if "k_pyr_base" in r[0]]
i0, i1 = bases[-3], bases[-2]
t0 = rows[i0][1]
lastend = {}
for name, s, e, q in rows[i0:i1 + 12]:
    gap = (s - lastend[q]) / 1000.0 if q in lastend else 0.0
    lastend[q] = e
    print(f"q{q:<3} +{(s - t0) / 1000.0:9.1f} us  dur {(e - s) / 1000.0:8.1f} us  gap {gap:7.1f}  {name.split('(')[0][:40]}")
print("columns:", cols)
