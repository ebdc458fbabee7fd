#!/bin/bash
mkdir -p gpurun_out/r03j
export TMPDIR=/tmp
timeout 400 python3 -m pytest tests/test_gpu_extractor.py tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/r03j/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r03j/pytest.log
bash tools/ab.sh "ORBX_FAST_SPLIT=0" "ORBX_NONE=1" 2>&1 | tee gpurun_out/r03j/ab.log
for wl in tumvi kitti; do for v in "ORBX_FAST_SPLIT=0" "ORBX_NONE=1"; do
  r=$(env $v python3 bench.py --workload $wl --steps 30 --warmup 5 --cpu-frames 0 --no-profile --verify 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$wl $v : $r" | tee -a gpurun_out/r03j/ab.log
done; done
bash tools/timeline.sh > gpurun_out/r03j/timeline_split.txt 2>&1; head -36 gpurun_out/r03j/timeline_split.txt
