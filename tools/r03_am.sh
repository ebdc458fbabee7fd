#!/bin/bash
# visit am: k_compact's gather inside the first quad-tree tier (one launch less on the main stream's latency-bound stretch)
export TMPDIR=/tmp
mkdir -p gpurun_out/r03am
run() { wl=$1; shift; env "$@" timeout 200 python bench.py --workload $wl --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']; print('$wl $*', d['value'], d['ms_per_step'], r['median'], r['min'], r['max'], d.get('parity_checked'))"; }
{
timeout 600 python -m pytest tests/test_gpu_extractor.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -2
for wl in euroc kitti tumvi; do run $wl ORBX_COMPACT_LAUNCH=1; run $wl ORBX_NONE=1; run $wl ORBX_COMPACT_LAUNCH=1; run $wl ORBX_NONE=1; done
for v in 1 0; do
  ORBX_COMPACT_LAUNCH=$v ORBX_SIDE_STREAMS=0 timeout 90 rocprofv3 --kernel-trace --stats -d gpurun_out/r03am/se$v -o se -- python bench.py --pmc-child --workload euroc --steps 12 --warmup 3 > /dev/null 2>&1
  python tools/rocprof_summary.py $(find gpurun_out/r03am/se$v -name "*.db" | head -1) gpurun_out/r03am/serialized_$v.csv | grep -i "compact\|octree"
  rm -rf gpurun_out/r03am/se$v
done
} > gpurun_out/r03am/log.txt 2>&1
cat gpurun_out/r03am/log.txt
