#!/bin/bash
# visit x: stereo Hamming / SAD stages with 16 lanes per left keypoint on the hardware (tests, KITTI line, timeline); batch sizes of the other workloads
mkdir -p gpurun_out/r03x
run() { wl=$1; shift; timeout 200 python bench.py --workload $wl --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']; print('$wl $*', d['value'], d['ms_per_step'], r['median'], r['min'], r['max'], d.get('parity_checked'))"; }
{
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_matcher.py tests/test_gpu_adapter_vs_reference.py -x -q -m gpu -k "stereo" 2>&1 | tail -3
run kitti; run kitti
run kitti --batch 128
run tumvi; run tumvi --batch 128; run tumvi --batch 256
bash tools/workload_timeline.sh kitti
} > gpurun_out/r03x/log.txt 2>&1
cat gpurun_out/r03x/log.txt | head -150
