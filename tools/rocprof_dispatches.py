#!/usr/bin/env python3
"""Per-(kernel, grid) average duration from a rocprofv3 results .db (view kernels): separates the launches of one kernel
that differ by problem size, e.g. the seven pyramid levels of k_pyr_resize."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
try:
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    q = "select name, grid_size_x, grid_size_y, count(*), avg(end - start) / 1000.0 from kernels group by name, grid_size_x, grid_size_y order by name, grid_size_x desc"
    if "grid_size_x" not in cols:
        q = "select name, grid_x, grid_y, count(*), avg(end - start) / 1000.0 from kernels group by name, grid_x, grid_y order by name, grid_x desc"
    for name, gx, gy, n, avg in c.execute(q):
        print(f"{name.split('(')[0]:40s} grid {gx:>9} x {gy:<5} calls {n:4d}  avg {avg:9.2f} us")
except Exception as e:
    print("schema:", e, [r[0] for r in c.execute("select name from sqlite_master")][:60])
    print(cols)
