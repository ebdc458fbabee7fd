#!/usr/bin/env python3
"""Kernel timeline of the host-pointer matcher calls: run tools/latency_calls.py under rocprofv3 --kernel-trace, then for every entry point print
the dispatches of ONE steady-state call (the last but one) with start offset, duration and the gap to the previous dispatch.
usage: python tools/call_timeline.py <results.db>"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [t[0] for t in c.execute("select name from sqlite_master where type in ('table','view')")]
kview = "kernels" if "kernels" in tabs else next(t for t in tabs if t.startswith("kernels"))
rows = [(re.sub(r"<.*>", "", n.split("(")[0].replace("orbx::", "").replace("void ", "")), s, e) for n, s, e in c.execute(f"select name, start, end from {kview} order by start")]
# a "call" = a maximal run of dispatches separated by gaps < 40 us that starts with k_xfer
calls, cur = [], []
for r in rows:
    if cur and (r[1] - cur[-1][2] > 40000 or (r[0] == "k_xfer" and cur[-1][0] != "k_xfer" and r[1] - cur[-1][2] > 8000)):
        calls.append(cur)
        cur = []
    cur.append(r)
if cur:
    calls.append(cur)
sig = {}
for cl in calls:
    key = tuple(r[0] for r in cl)
    sig.setdefault(key, []).append(cl)
for key, cls in sig.items():
    if len(cls) < 5 or "k_xfer" not in key:
        continue
    cl = cls[-2]
    t0 = cl[0][1]
    print(f"--- {len(cls)} calls of: {' > '.join(key)}   (first start -> last end {(cl[-1][2] - t0) / 1e3:.1f} us)")
    prev = t0
    for n, s, e in cl:
        print(f"   +{(s - t0) / 1e3:7.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:6.1f}  {n}")
        prev = e
    import statistics
    for i, n in enumerate(key):
        print(f"      median dur {n}: {statistics.median((c_[i][2] - c_[i][1]) / 1e3 for c_ in cls):.1f} us")
