#!/bin/bash
# visit ad: levels 1..7 of the pyramid in one launch with in-kernel row-block dependencies (k_pyr_resize_chain_ordered) against one launch per level
export TMPDIR=/tmp
mkdir -p gpurun_out/r03ad
run() { wl=$1; shift; env "$@" timeout 200 python bench.py --workload $wl --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']; print('$wl $*', d['value'], d['ms_per_step'], r['median'], r['min'], r['max'], d.get('parity_checked'))"; }
{
timeout 300 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -2
for wl in euroc kitti tumvi; do
  run $wl ORBX_PYR_CHAIN=0; run $wl ORBX_NONE=1; run $wl ORBX_PYR_CHAIN=0; run $wl ORBX_NONE=1
done
for v in 0 1; do
  ORBX_PYR_CHAIN=$v ORBX_SIDE_STREAMS=0 timeout 90 rocprofv3 --kernel-trace --stats -d gpurun_out/r03ad/se$v -o se -- python bench.py --pmc-child --workload euroc --steps 12 --warmup 3 > /dev/null 2>&1
  python tools/rocprof_summary.py $(find gpurun_out/r03ad/se$v -name "*.db" | head -1) gpurun_out/r03ad/serialized_chain$v.csv | grep -i "pyr"
  rm -rf gpurun_out/r03ad/se$v
done
} > gpurun_out/r03ad/log.txt 2>&1
cat gpurun_out/r03ad/log.txt
