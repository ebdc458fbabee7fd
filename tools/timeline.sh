#!/bin/bash
# kernel + copy timeline of the PIPELINED bench loop (side streams on): every dispatch of one steady-state step with its queue, start offset and
# duration, the busy union and the idle gaps of the main queue.  usage: bash tools/timeline.sh [ENV=VALUE ...]
export TMPDIR=/tmp
O=gpurun_out/tl; rm -rf $O; mkdir -p $O
env WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 "$@" timeout 120 rocprofv3 --kernel-trace ${TL_COPIES:+--memory-copy-trace} -d $O -o s -- python bench.py --steps 8 --warmup 3 --settle 4 --settle-seconds 0 --cpu-frames 0 --no-profile --verify 0 --repeat 1 --latency 0 --no-pmc --no-other-workloads ${TL_ARGS:-} > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import sqlite3,glob,re
db=sorted(glob.glob('gpurun_out/tl/**/*.db',recursive=True))[-1]
c=sqlite3.connect(db)
tabs=[t[0] for t in c.execute("select name from sqlite_master where type in ('table','view')")]
rows=[(re.sub(r"\(.*","",n).replace("orbx::","").replace("void ",""),q,s,e) for n,q,s,e in c.execute("select name,queue_id,start,end from kernels order by start")]
cp=[]
if 'memory_copies' in tabs:
    cols=[r[1] for r in c.execute("pragma table_info(memory_copies)")]
    try: cp=[("copy:"+str(n),-1,s,e) for n,s,e in c.execute("select name,start,end from memory_copies order by start")]
    except Exception as ex: print("copies:",ex,cols)
ib=[i for i,r in enumerate(rows) if 'k_pyr_stream' in r[0] or 'k_pyr_base' in r[0]]
# settle 4 + warm-up 3 + 8 timed steps with resident input, then warm-up 3 + 8 with host input: steps 10 and 11 are steady-state resident ones
k=int(__import__('os').environ.get('TL_STEP','10'))
i0,i1=ib[k],ib[k+2]
t0,t1=rows[i0][2],rows[i1][2]
print("step length us", (t1-t0)/1000.0)
ev=sorted([r for r in rows if t0<=r[2]<t1]+[r for r in cp if t0<=r[2]<t1], key=lambda r:r[2])
for n,q,s,e in ev: print(f"{(s-t0)/1000.0:9.1f} {(e-s)/1000.0:8.1f}  q{q}  {n[:48]}")
iv=sorted((s,e) for n,q,s,e in ev if q!=-1)
tot=0; cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce: tot+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
tot+=ce-cs
print("busy (union of kernels) us", tot/1000.0, " sum of kernel durations us", sum(e-s for s,e in iv)/1000.0)
PY
