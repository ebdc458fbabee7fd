#!/bin/bash
# visit ag: FAST of level 0 beside the one-launch chain on the spare / input stream (the aux stream shares a hardware queue with the matcher)
export TMPDIR=/tmp
mkdir -p gpurun_out/r03ag
run() { wl=$1; shift; env "$@" timeout 200 python bench.py --workload $wl --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 $EXTRA 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']; print('$wl $* $EXTRA', d['value'], d['ms_per_step'], r['median'], r['min'], r['max'], d.get('parity_checked'))"; }
{
EXTRA=""
run euroc ORBX_FAST_SPLIT=0; run euroc ORBX_FAST_SPLIT_STREAM=1; run euroc ORBX_FAST_SPLIT_STREAM=2; run euroc ORBX_FAST_SPLIT=0; run euroc ORBX_FAST_SPLIT_STREAM=1; run euroc ORBX_FAST_SPLIT_STREAM=2
bash tools/timeline.sh ORBX_FAST_SPLIT_STREAM=1 | head -28
bash tools/timeline.sh ORBX_FAST_SPLIT_STREAM=2 | head -28
} > gpurun_out/r03ag/log.txt 2>&1
cat gpurun_out/r03ag/log.txt
