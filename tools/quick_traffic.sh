#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/qt; rm -rf $O; mkdir -p $O
bash tools/quick_prof.sh euroc
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pf -o pf -- python bench.py --pmc-child --workload euroc --steps 2 --warmup 1 > /dev/null 2>&1
python3 - <<'PY'
import sqlite3,glob,re
from collections import defaultdict
db=sorted(glob.glob('gpurun_out/qt/pf/**/*.db',recursive=True))[-1]
c=sqlite3.connect(db); acc=defaultdict(lambda:[0,0])
for name,val in c.execute("select kernel_name, value from counters_collection where counter_name='FETCH_SIZE'"):
    k=re.sub(r"<.*>","",name.split("(")[0].replace("void ","").replace("orbx::","")); acc[k][0]+=val; acc[k][1]+=1
for k,v in sorted(acc.items()):
    if k.startswith('k_'): print(k, round(v[0]/v[1]*1024/1e6,1), "MB fetched per launch")
PY
