#!/bin/bash
# round 4, GPU visit 1: OpenCV probe, GPU suite, matcher-deferral A/B, two ranks on one GPU, TUM-VI / KITTI at 128 per step
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04v1; mkdir -p $O
{
echo "== OpenCV probe"; python3 -c "import cv2; print('cv2', cv2.__version__)" 2>&1 | tail -1
timeout 40 pip install opencv-python-headless 2>&1 | tail -2
timeout 40 pip download opencv-python-headless -d /tmp/w 2>&1 | tail -1
find / -xdev \( -name "libopencv_core*" -o -name "cv2*.so" -o -name "opencv4" \) 2>/dev/null | head -5; echo "(end of find)"
ls /usr/include/opencv4 /usr/local/include/opencv4 2>&1 | head -3
} > $O/opencv_probe.log 2>&1
cat $O/opencv_probe.log
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
bash tools/ab.sh -t r04v1 -r 2 "ORBX_NONE=1" "ORBX_MATCH_DEFER=1" "ORBX_MATCH_DEFER=2" > /dev/null 2>&1; cat $O/ab.log
for v in ORBX_NONE=1 ORBX_MATCH_DEFER=1 ORBX_MATCH_DEFER=2; do echo "== timeline $v"; bash tools/timeline.sh $v 2>&1 | head -40; done > $O/timelines.txt 2>&1
grep -A3 "== timeline\|step length\|busy" $O/timelines.txt | head -40
timeout 400 python3 bench.py --gpus 2 --share-gpus --steps 20 --warmup 5 --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 --no-profile 2> $O/two_ranks.err | tail -1 > $O/two_ranks_one_gpu.json; echo "two ranks rc=$?"; cut -c1-600 $O/two_ranks_one_gpu.json; tail -3 $O/two_ranks.err
for wl in tumvi kitti; do timeout 300 python3 bench.py --workload $wl --cpu-frames 0 --no-pmc 2> $O/bench_$wl.err | tail -1 > $O/bench_$wl.json; echo "$wl rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_$wl.json')); print(d['value'], d['ms_per_step'], d['repeats'], d.get('parity_checked'))"; done
