#!/bin/bash
# kernel timeline of the steady-state steps with the side streams ON: start offset, duration and queue of every dispatch of the last step
export TMPDIR=/tmp
O=gpurun_out/tl; rm -rf $O; mkdir -p $O
env "$@" timeout 60 rocprofv3 --kernel-trace -d $O -o s -- python bench.py --pmc-child --workload ${WL:-euroc} --steps 6 --warmup 2 > /dev/null 2>&1
python3 - <<'PY'
import sqlite3,glob,re
db=sorted(glob.glob('gpurun_out/tl/**/*.db',recursive=True))[-1]
c=sqlite3.connect(db)
rows=list(c.execute("select name,queue_id,start,end from kernels order by start"))
# last step: from the last k_pyr_base dispatch on
ib=[i for i,r in enumerate(rows) if 'k_pyr_base' in r[0]]
i0,i1=ib[-2],ib[-1]
t0=rows[i0][2]
print("step length us", (rows[i1][2]-t0)/1000.0)
busy=0; cur_end=t0
for name,q,s,e in rows[i0:i1]:
    n=re.sub(r"\(.*","",name).replace("orbx::","").replace("void ","")
    print(f"{(s-t0)/1000.0:9.1f} {(e-s)/1000.0:8.1f}  q{q}  {n}")
# union of busy intervals inside the step
iv=sorted((s,e) for _,_,s,e in rows[i0:i1])
tot=0; cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce: tot+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
tot+=ce-cs
print("busy (union) us", tot/1000.0, " sum of durations us", sum(e-s for s,e in iv)/1000.0)
PY
