#!/bin/bash
# dynamic instruction counts of k_fast_ini per phase, by truncation (ORBX_FAST_STOP=1..4, 0 = whole kernel)
export TMPDIR=/tmp
for st in 1 2 3 4 0; do
  O=gpurun_out/fv$st; rm -rf $O; mkdir -p $O
  ORBX_FAST_STOP=$st ORBX_SIDE_STREAMS=0 timeout 60 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d $O -o s -- python bench.py --pmc-child --workload euroc --steps 2 --warmup 1 > /dev/null 2>&1
  python3 - $st <<'PY'
import sqlite3,glob,sys
from collections import defaultdict
dbs=sorted(glob.glob(f'gpurun_out/fv{sys.argv[1]}/**/*.db',recursive=True))
if not dbs: print('stop',sys.argv[1],'no db'); sys.exit(0)
c=sqlite3.connect(dbs[-1]); acc=defaultdict(lambda:[0.0,0])
for name,cn,val in c.execute("select kernel_name, counter_name, value from counters_collection"):
    if 'k_fast_ini' in name: acc[cn][0]+=val; acc[cn][1]+=1
w=acc['SQ_WAVES'][0]/max(acc['SQ_WAVES'][1],1)
print('stop',sys.argv[1],'waves',int(w),' per wave:',{k.replace('SQ_INSTS_',''):round(v[0]/v[1]/w,1) for k,v in sorted(acc.items()) if k!='SQ_WAVES'})
PY
done
