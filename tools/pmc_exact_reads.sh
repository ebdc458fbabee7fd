#!/bin/bash
# exact memory-side read bytes per kernel from the request-size classes: 32*RDREQ_32B + 64*RDREQ_64B + 128*RDREQ_128B (calibrates FETCH_SIZE)
export TMPDIR=/tmp
O=gpurun_out/exact; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum -d $O/p -o p -- python bench.py --pmc-child --workload euroc --steps 2 --warmup 1 > /dev/null 2>&1
python3 - <<'PY'
import sqlite3,glob,re
from collections import defaultdict
db=sorted(glob.glob('gpurun_out/exact/p/**/*.db',recursive=True))[-1]
c=sqlite3.connect(db); acc=defaultdict(lambda: defaultdict(lambda:[0,0]))
for name,cn,val in c.execute("select kernel_name, counter_name, value from counters_collection"):
    k=re.sub(r"<.*>","",name.split("(")[0].replace("void ","").replace("orbx::","")); a=acc[k][cn]; a[0]+=val; a[1]+=1
print("kernel, req32, req64, req128, req_all, exact_read_MB")
for k,v in sorted(acc.items()):
    if not k.startswith('k_'): continue
    g=lambda n: v[n][0]/max(v[n][1],1)
    r32,r64,r128,ra=g('TCC_EA0_RDREQ_32B_sum'),g('TCC_EA0_RDREQ_64B_sum'),g('TCC_EA0_RDREQ_128B_sum'),g('TCC_EA0_RDREQ_sum')
    print(k, int(r32), int(r64), int(r128), int(ra), round((32*r32+64*r64+128*r128)/1e6,1))
PY
