#!/bin/bash
# one GPU-box visit: GPU test-suite + the three bench workloads; outputs under gpurun_out/round/
mkdir -p gpurun_out/round
timeout 1200 python3 -m pytest tests -m gpu -x -q > gpurun_out/round/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/round/pytest.log
for wl in euroc kitti tumvi; do
  timeout 600 python3 bench.py --workload $wl "$@" > gpurun_out/round/bench_$wl.json 2> gpurun_out/round/bench_$wl.err; echo "bench $wl rc=$?"
  tail -c 1500 gpurun_out/round/bench_$wl.err; cut -c1-900 gpurun_out/round/bench_$wl.json
done
# the OpenCV pin: the first box that has cv2 generates the goldens tests/test_oracle_vs_opencv.py waits for
python3 -c "import cv2; print('cv2', cv2.__version__)" 2>/dev/null && python3 tools/gen_ocv_golden.py && echo "OpenCV goldens written: commit tests/golden/ocv_*.npz" || echo "cv2: not importable on this box"
