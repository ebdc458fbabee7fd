#!/bin/bash
# k_fast_ini duration by cells (waves) per workgroup (ORBX_FAST_INI_WAVES) and, for the default, with tile loads only (ORBX_FAST_STOP=1)
export TMPDIR=/tmp
run() {  # tag, env...
  local tag=$1; shift
  local O=gpurun_out/fw_$tag; rm -rf $O; mkdir -p $O
  env "$@" ORBX_SIDE_STREAMS=0 timeout 60 rocprofv3 --kernel-trace --stats -d $O -o s -- python bench.py --pmc-child --workload euroc --steps 3 --warmup 1 > /dev/null 2>&1
  python3 - $tag <<'PY'
import sqlite3,glob,sys
dbs=sorted(glob.glob(f'gpurun_out/fw_{sys.argv[1]}/**/*.db',recursive=True))
if not dbs: print(sys.argv[1],'no db'); sys.exit(0)
c=sqlite3.connect(dbs[-1])
for name,calls,tot,avg,pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if 'k_fast' in name: print(sys.argv[1], name[:40], "calls", calls, "avg us", round(avg/1000 if avg>10000 else avg,1))
PY
}
for w in 1 2 4 8; do run w$w ORBX_FAST_INI_WAVES=$w; done
run w4_stop1 ORBX_FAST_INI_WAVES=4 ORBX_FAST_STOP=1
run w8_stop1 ORBX_FAST_INI_WAVES=8 ORBX_FAST_STOP=1
