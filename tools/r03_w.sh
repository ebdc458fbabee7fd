#!/bin/bash
# visit w: A/B of three scheduling switches on EuRoC and TUM-VI: blur on another stream (hardware-queue sharing with the matcher), a matcher problem's
# workgroups on one XCD, the previous batch's downloads issued behind the next batch's resize chain
mkdir -p gpurun_out/r03w
run() { wl=$1; shift; env "$@" timeout 200 python bench.py --workload $wl --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']; print('$wl $*', d['value'], d['ms_per_step'], r['median'], r['min'], r['max'], d.get('parity_checked'))"; }
{
for wl in euroc tumvi; do
  run $wl ORBX_NONE=1
  run $wl ORBX_BLUR_STREAM=1
  run $wl ORBX_BLUR_STREAM=2
  run $wl ORBX_MATCH_XCD=1
  run $wl ORBX_COPY_DEFER=1
  run $wl ORBX_COPY_DEFER=1 ORBX_MATCH_XCD=1 ORBX_BLUR_STREAM=1
  run $wl ORBX_NONE=1
done
bash tools/workload_timeline.sh tumvi ORBX_BLUR_STREAM=1
bash tools/workload_timeline.sh euroc ORBX_COPY_DEFER=1
} > gpurun_out/r03w/log.txt 2>&1
cat gpurun_out/r03w/log.txt | head -150
