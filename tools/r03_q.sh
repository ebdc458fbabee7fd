#!/bin/bash
mkdir -p gpurun_out/r03q
export TMPDIR=/tmp
timeout 400 python3 -m pytest tests/test_gpu_extractor.py tests/test_gpu_adapter_vs_reference.py -m gpu -x -q > gpurun_out/r03q/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r03q/pytest.log
bash tools/ab.sh "ORBX_SECOND_PASS=list" "ORBX_NONE=1" 2>&1 | tee gpurun_out/r03q/ab.log
for v in "ORBX_SECOND_PASS=list" "ORBX_NONE=1"; do for wl in kitti tumvi tumvi; do
  r=$(env $v python3 bench.py --workload $wl --steps 30 --warmup 5 --cpu-frames 0 --no-profile --verify 0 --repeat 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$wl $v : $r" | tee -a gpurun_out/r03q/ab.log
done; done
bash tools/quick_prof.sh euroc > /dev/null 2>&1; grep "strip\|list" gpurun_out/qp/stats.csv | tee -a gpurun_out/r03q/ab.log
