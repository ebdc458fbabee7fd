#!/bin/bash
# kernel timeline of ONE synchronous single-frame orbx_extract call (tools/latency.py under rocprofv3 --kernel-trace): every dispatch with its
# queue, start offset and duration, and the gaps between them.  usage: bash tools/latency_timeline.sh [ENV=VALUE ...]
export TMPDIR=/tmp
O=gpurun_out/lt; rm -rf $O; mkdir -p $O
env "$@" timeout 120 rocprofv3 --kernel-trace -d $O -o s -- python tools/latency.py > $O/latency.txt 2> $O/latency.err
cat $O/latency.txt
python3 - <<'PY'
import sqlite3,glob,re
db=sorted(glob.glob('gpurun_out/lt/**/*.db',recursive=True))[-1]
c=sqlite3.connect(db)
rows=[(re.sub(r"\(.*","",n).replace("orbx::","").replace("void ",""),q,s,e) for n,q,s,e in c.execute("select name,queue_id,start,end from kernels order by start")]
ib=[i for i,r in enumerate(rows) if 'k_pyr_base' in r[0]]
k=len(ib)//2
i0,i1=ib[k],ib[k+1]
t0=rows[i0][2]
print("call-to-call us", (rows[i1][2]-t0)/1000.0, " first kernel start -> last kernel end us", (max(r[3] for r in rows[i0:i1])-t0)/1000.0)
prev=t0
for n,q,s,e in rows[i0:i1]:
    print(f"{(s-t0)/1000.0:9.1f} {(e-s)/1000.0:8.1f}  gap {(s-prev)/1000.0:7.1f}  q{q}  {n[:48]}")
    prev=max(prev,e)
print("sum of kernel durations us", sum(e-s for n,q,s,e in rows[i0:i1])/1000.0)
PY
