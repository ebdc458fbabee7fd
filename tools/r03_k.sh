#!/bin/bash
mkdir -p gpurun_out/r03k
export TMPDIR=/tmp
bash tools/ab.sh "GPU_MAX_HW_QUEUES=8 ORBX_FAST_SPLIT=0" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=6" 2>&1 | tee gpurun_out/r03k/ab.log
for wl in tumvi kitti; do for v in "GPU_MAX_HW_QUEUES=8 ORBX_FAST_SPLIT=0" "GPU_MAX_HW_QUEUES=8"; do
  r=$(env $v python3 bench.py --workload $wl --steps 30 --warmup 5 --cpu-frames 0 --no-profile --verify 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$wl $v : $r" | tee -a gpurun_out/r03k/ab.log
done; done
GPU_MAX_HW_QUEUES=8 bash tools/timeline.sh > gpurun_out/r03k/timeline_split_q8.txt 2>&1; head -34 gpurun_out/r03k/timeline_split_q8.txt
