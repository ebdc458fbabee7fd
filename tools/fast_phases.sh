#!/bin/bash
# per-phase cost of k_fast_ini by truncation (ORBX_FAST_STOP=1..4: return after tile load / stage A / stage B / exact scores)
export TMPDIR=/tmp
for st in 1 2 3 4 0; do
  O=gpurun_out/fp$st; rm -rf $O; mkdir -p $O
  ORBX_FAST_STOP=$st ORBX_SIDE_STREAMS=0 timeout 60 rocprofv3 --kernel-trace --stats -d $O -o s -- python bench.py --pmc-child --workload euroc --steps 3 --warmup 1 > /dev/null 2>&1
  python3 - $st <<'PY'
import sqlite3,glob,sys
dbs=sorted(glob.glob(f'gpurun_out/fp{sys.argv[1]}/**/*.db',recursive=True))
if not dbs: print('stop',sys.argv[1],'no db'); sys.exit(0)
db=dbs[-1]
c=sqlite3.connect(db)
for name,calls,tot,avg,pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if 'k_fast_ini' in name: print("stop",sys.argv[1],"k_fast_ini avg us", round(avg/1000 if avg>10000 else avg,1))
PY
done
