#!/bin/bash
# round 4, GPU visit 3: k_blur_stream (throttled blur beside the FAST strips) against k_blur_pk
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04v3; mkdir -p $O
ORBX_BLUR_STREAM=1024 timeout 900 python3 -m pytest tests/test_gpu_extractor.py tests/test_gpu_pipeline.py -m gpu -x -q > $O/pytest_stream.log 2>&1; echo "pytest (stream 1024) rc=$?"; tail -3 $O/pytest_stream.log
ORBX_BLUR_STREAM=96 timeout 900 python3 -m pytest tests/test_gpu_extractor.py -m gpu -x -q -k "not alternative" > $O/pytest_stream96.log 2>&1; echo "pytest (stream 96) rc=$?"; tail -2 $O/pytest_stream96.log
bash tools/ab.sh -t r04v3 -r 2 -p "ORBX_NONE=1" "ORBX_BLUR_STREAM=1024" "ORBX_BLUR_STREAM=512" "ORBX_BLUR_STREAM=2048" "ORBX_BLUR_STREAM=4096" > /dev/null 2>&1; cat $O/ab.log
for v in ORBX_BLUR_STREAM=1024; do echo "== timeline $v"; bash tools/timeline.sh $v 2>&1 | head -44; done > $O/timelines.txt 2>&1
bash tools/ab.sh -t r04v3b -r 1 -w "kitti tumvi" "ORBX_NONE=1" "ORBX_BLUR_STREAM=1024" "ORBX_BLUR_STREAM=2048" > /dev/null 2>&1; cat gpurun_out/r04v3b/ab.log
