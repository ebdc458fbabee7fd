#!/bin/bash
mkdir -p gpurun_out/r03m
export TMPDIR=/tmp
bash tools/ab.sh "ORBX_NONE=1" "ORBX_STRIP_WAVES=8" 2>&1 | tee gpurun_out/r03m/ab.log
for v in "ORBX_NONE=1" "ORBX_STRIP_WAVES=8"; do for wl in tumvi tumvi tumvi kitti; do
  r=$(env $v python3 bench.py --workload $wl --steps 30 --warmup 5 --cpu-frames 0 --no-profile --verify 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$wl $v : $r" | tee -a gpurun_out/r03m/ab.log
done; done
ORBX_STRIP_WAVES=8 bash tools/quick_prof.sh euroc > /dev/null 2>&1; grep "strip\|list" gpurun_out/qp/stats.csv | tee -a gpurun_out/r03m/ab.log
