#!/bin/bash
# visit af: FAST of level 0 on the aux stream beside the ONE-launch pyramid chain (was level / worse beside seven launches)
export TMPDIR=/tmp
mkdir -p gpurun_out/r03af
run() { wl=$1; shift; env "$@" timeout 200 python bench.py --workload $wl --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 $EXTRA 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']; print('$wl $* $EXTRA', d['value'], d['ms_per_step'], r['median'], r['min'], r['max'], d.get('parity_checked'))"; }
{
EXTRA=""
run euroc ORBX_FAST_SPLIT=0; run euroc ORBX_NONE=1; run euroc ORBX_FAST_SPLIT=0; run euroc ORBX_NONE=1
EXTRA="--batch 128"
for wl in kitti tumvi; do run $wl ORBX_FAST_SPLIT=0; run $wl ORBX_NONE=1; done
bash tools/timeline.sh | head -60
} > gpurun_out/r03af/log.txt 2>&1
cat gpurun_out/r03af/log.txt
