#!/bin/bash
# visit ae: the one-launch pyramid chain at 128 frames per step (threshold of the default), defaults of the three workloads, the GPU suite
export TMPDIR=/tmp
mkdir -p gpurun_out/r03ae
run() { wl=$1; shift; env "$@" timeout 200 python bench.py --workload $wl --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 $EXTRA 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']; print('$wl $* $EXTRA', d['value'], d['ms_per_step'], r['median'], r['min'], r['max'], d.get('parity_checked'))"; }
{
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
EXTRA="--batch 128"
for wl in euroc kitti tumvi; do run $wl ORBX_PYR_CHAIN=0; run $wl ORBX_NONE=1; run $wl ORBX_PYR_CHAIN=0; run $wl ORBX_NONE=1; done
EXTRA=""
for wl in euroc kitti tumvi; do run $wl ORBX_NONE=1; done
} > gpurun_out/r03ae/log.txt 2>&1
cat gpurun_out/r03ae/log.txt
