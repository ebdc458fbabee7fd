#!/bin/bash
# visit v: the stereo row index + pipelined stereo downloads on the hardware: the stereo GPU tests, the KITTI bench line, the KITTI step's timeline
mkdir -p gpurun_out/r03v
{
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_matcher.py -x -q -m gpu -k "stereo" 2>&1 | tail -3
for i in 1 2; do timeout 200 python bench.py --workload kitti --cpu-frames 0 --no-pmc 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('kitti', d['value'], d['ms_per_step'], d.get('repeats'), d.get('parity_checked'))"; done
bash tools/workload_timeline.sh kitti
} > gpurun_out/r03v/log.txt 2>&1
cat gpurun_out/r03v/log.txt | head -120
