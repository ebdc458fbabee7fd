#!/bin/bash
# branch r04-matcher-beside-fast: A/B of ORBX_MATCH_DEFER=1 (the previous batch's matcher + match-vector copy issued behind the next batch's pyramid chain
# and blur launch, beside its FAST strips) against the default (issued at once: beside the next batch's pyramid).  usage: gpurun -- 'bash tools/r04_match_defer_ab.sh'
export TMPDIR=/tmp
mkdir -p gpurun_out/r04a
run() { wl=$1; shift; env "$@" timeout 200 python bench.py --workload $wl --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']; print('$wl $*', d['value'], d['ms_per_step'], r['median'], r['min'], r['max'], d.get('parity_checked'))"; }
{
timeout 300 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -1
ORBX_MATCH_DEFER=1 timeout 300 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -1
run euroc ORBX_NONE=1; run euroc ORBX_MATCH_DEFER=1; run euroc ORBX_NONE=1; run euroc ORBX_MATCH_DEFER=1
bash tools/timeline.sh ORBX_MATCH_DEFER=1 | head -30
} > gpurun_out/r04a/log.txt 2>&1
cat gpurun_out/r04a/log.txt
