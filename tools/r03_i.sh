#!/bin/bash
mkdir -p gpurun_out/r03i
export TMPDIR=/tmp
timeout 300 python3 -m pytest tests/test_gpu_extractor.py -m gpu -x -q > gpurun_out/r03i/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r03i/pytest.log
bash tools/ab.sh "ORBX_NONE=1" 2>&1 | tee gpurun_out/r03i/ab.log
for wl in tumvi kitti; do
  r=$(python3 bench.py --workload $wl --steps 30 --warmup 5 --cpu-frames 0 --no-profile --verify 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$wl : $r" | tee -a gpurun_out/r03i/ab.log
done
bash tools/quick_prof.sh euroc sq > /dev/null 2>&1; cp gpurun_out/qp/stats.csv gpurun_out/r03i/stats.csv; cp gpurun_out/qp/qp_pmc_sq_counters.csv gpurun_out/r03i/sq.csv; head -16 gpurun_out/r03i/stats.csv; grep "strip\|describe\|march" gpurun_out/r03i/sq.csv
