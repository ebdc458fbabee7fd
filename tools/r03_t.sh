#!/bin/bash
# visit t: the new tests on the hardware (single-frame host block, fisheye stereo member), single-frame latency with the final settings
mkdir -p gpurun_out/r03t
{
timeout 900 python -m pytest tests/test_gpu_extractor.py tests/test_gpu_adapter_vs_reference.py tests/test_gpu_matcher.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do python tools/latency.py 2>/dev/null; done
bash tools/latency_timeline.sh ORBX_NONE=1
python bench.py --steps 20 --warmup 5 --no-pmc --cpu-frames 0 --no-other-workloads 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('euroc', d['value'], d['ms_per_step'], d.get('latency'))"
} > gpurun_out/r03t/log.txt 2>&1
cat gpurun_out/r03t/log.txt
