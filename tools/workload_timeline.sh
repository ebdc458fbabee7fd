#!/bin/bash
# kernel timeline of one steady-state step of a bench workload (pipelined loop, side streams on) + per-kernel totals of that step
# usage: bash tools/workload_timeline.sh kitti|tumvi|euroc [ENV=VALUE ...]
export TMPDIR=/tmp
WL=$1; shift
O=gpurun_out/wt_$WL; rm -rf $O; mkdir -p $O
env WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 "$@" timeout 200 rocprofv3 --kernel-trace -d $O -o s -- python bench.py --workload $WL --steps 8 --warmup 3 --settle 4 --cpu-frames 0 --no-profile --verify 0 --repeat 1 --latency 0 --no-pmc --no-other-workloads > $O/bench.json 2> $O/bench.err
python3 - $O <<'PY'
import sqlite3,glob,re,sys,collections
db=sorted(glob.glob(sys.argv[1]+'/**/*.db',recursive=True))[-1]
c=sqlite3.connect(db)
rows=[(re.sub(r"\(.*","",n).replace("orbx::","").replace("void ",""),q,s,e) for n,q,s,e in c.execute("select name,queue_id,start,end from kernels order by start")]
ib=[i for i,r in enumerate(rows) if 'k_pyr_base' in r[0]]
# k_pyr_base launches per step: 1 (mono) or 2 (stereo rig); take a window of the resident-input loop: launches 3/8 of the way in
per=2 if 'kitti' in sys.argv[1] else 1
k=(4+3+4)*per
i0,i1=ib[k],ib[k+per]
t0,t1=rows[i0][2],rows[i1][2]
print("step length us", (t1-t0)/1000.0)
ev=[r for r in rows if t0<=r[2]<t1]
for n,q,s,e in ev: print(f"{(s-t0)/1000.0:9.1f} {(e-s)/1000.0:8.1f}  q{q}  {n[:52]}")
tot=collections.Counter(); cnt=collections.Counter()
for n,q,s,e in ev: tot[n]+=(e-s)/1000.0; cnt[n]+=1
print("per-kernel totals of the step (us, launches):")
for n,v in tot.most_common(): print(f"{v:9.1f} {cnt[n]:3d}  {n[:60]}")
print("sum of kernel durations us", sum(tot.values()))
PY
