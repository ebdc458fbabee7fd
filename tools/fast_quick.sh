#!/bin/bash
# k_fast_* durations (rocprofv3 --kernel-trace --stats, serialized streams) for the current build; extra env as arguments
export TMPDIR=/tmp
O=gpurun_out/fq; rm -rf $O; mkdir -p $O
env "$@" ORBX_SIDE_STREAMS=0 timeout 60 rocprofv3 --kernel-trace --stats -d $O -o s -- python bench.py --pmc-child --workload euroc --steps 6 --warmup 2 > /dev/null 2>&1
python3 - <<'PY'
import sqlite3,glob
dbs=sorted(glob.glob('gpurun_out/fq/**/*.db',recursive=True))
c=sqlite3.connect(dbs[-1])
for name,calls,tot,avg,pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if 'k_' in name: print(name[:44].ljust(44), "calls", calls, "avg us", round(avg/1000 if avg>10000 else avg,1))
PY
