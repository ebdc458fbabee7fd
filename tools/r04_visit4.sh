#!/bin/bash
# round 4, GPU visit 4: k_pyr_resize_march_blur (blur fused into the pyramid march) against the separate blur
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04v4; mkdir -p $O
timeout 1200 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bash tools/ab.sh -t r04v4 -r 2 -p "ORBX_BLUR_FUSED=0" "ORBX_NONE=1" > /dev/null 2>&1; cat $O/ab.log
bash tools/ab.sh -t r04v4b -r 1 -w "kitti tumvi" "ORBX_BLUR_FUSED=0" "ORBX_NONE=1" "ORBX_BLUR_STREAM=4096" > /dev/null 2>&1; cat gpurun_out/r04v4b/ab.log
for v in ORBX_NONE=1; do echo "== timeline $v"; bash tools/timeline.sh $v 2>&1 | head -44; done > $O/timelines.txt 2>&1
