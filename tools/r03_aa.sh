#!/bin/bash
# visit aa: map-point window scan with 4 lanes per query (TUM-VI workload), A/B against the 16-lane form
mkdir -p gpurun_out/r03aa
run() { wl=$1; shift; env "$@" timeout 200 python bench.py --workload $wl --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']; print('$wl $*', d['value'], d['ms_per_step'], r['median'], r['min'], r['max'], d.get('parity_checked'))"; }
{
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_matcher.py -x -q -m gpu -k "mappoint or projection" 2>&1 | tail -3
run tumvi ORBX_MP_LPQ=16; run tumvi ORBX_NONE=1; run tumvi ORBX_MP_LPQ=16; run tumvi ORBX_NONE=1
bash tools/workload_timeline.sh tumvi | grep -v "copyBuffer"
} > gpurun_out/r03aa/log.txt 2>&1
cat gpurun_out/r03aa/log.txt | head -150
