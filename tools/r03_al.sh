#!/bin/bash
# visit al: the blur behind the FAST strips (beside the list pass / quad-tree stage) instead of beside them, with the round's final kernels
export TMPDIR=/tmp
mkdir -p gpurun_out/r03al
run() { wl=$1; shift; env "$@" timeout 200 python bench.py --workload $wl --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']; print('$wl $*', d['value'], d['ms_per_step'], r['median'], r['min'], r['max'], d.get('parity_checked'))"; }
{
for wl in euroc kitti tumvi; do run $wl ORBX_NONE=1; run $wl ORBX_BLUR_AFTER_FAST=1; run $wl ORBX_NONE=1; run $wl ORBX_BLUR_AFTER_FAST=1; done
bash tools/timeline.sh ORBX_BLUR_AFTER_FAST=1 | head -24
} > gpurun_out/r03al/log.txt 2>&1
cat gpurun_out/r03al/log.txt
