#!/bin/bash
# round 4, GPU visit 6: cheap knobs with the new blur kernel -- issue priority of the FAST strips, hardware queues, blur waves per workload, frames per step
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
bash tools/ab.sh -t r04v6 -r 2 -V "ORBX_NONE=1" "ORBX_FAST_PRIO=1" "ORBX_FAST_PRIO=3" "GPU_MAX_HW_QUEUES=8" "ORBX_BLUR_STREAM=3072" > /dev/null 2>&1; cat gpurun_out/r04v6/ab.log
bash tools/ab.sh -t r04v6b -r 1 -V -w "kitti tumvi" "ORBX_NONE=1" "ORBX_FAST_PRIO=3" "ORBX_BLUR_STREAM=3072" "ORBX_BLUR_STREAM=1536" > /dev/null 2>&1; cat gpurun_out/r04v6b/ab.log
for b in 512; do python3 bench.py --batch $b --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 --no-profile 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('euroc batch $b', d['value'], d['ms_per_step'], d['repeats']['ms_per_step'], bool(d['parity_checked']))"; done
