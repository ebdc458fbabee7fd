#!/bin/bash
# round 4, GPU visit 2: what the step is made of -- the pipelined EuRoC loop with parts left out (bench.py --ablate, diagnostic)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04v2; mkdir -p $O
run() { python3 bench.py --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 --no-profile "$@" 2>$O/err.txt | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']
print('$*', '|', d['value'], '| ms', d['ms_per_step'], 'median', r['median'], 'min', r['min'], 'max', r['max'], '| host-input ms', d['pcie_inclusive']['ms_per_step'])"; }
{
for rep in 1 2; do
run --ablate ""
run --ablate nodl
run --ablate nomatch
run --ablate nodl,nomatch
done
ORBX_SIDE_STREAMS=0 run --ablate ""
ORBX_SIDE_STREAMS=0 run --ablate nodl,nomatch
} 2>&1 | tee $O/ablate.log
for ab in nodl nomatch nodl,nomatch; do echo "== timeline --ablate $ab"; TL_ARGS="--ablate $ab" bash tools/timeline.sh 2>&1 | head -45; done > $O/timelines.txt 2>&1
grep "== timeline\|step length\|busy" $O/timelines.txt
