#!/bin/bash
# GPU visit: small-level chain (on / off), two-pixel corner score (on / off) on EuRoC and TUM-VI, new GPU tests, profile of the winner set
mkdir -p gpurun_out/r03e
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_extractor.py tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/r03e/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r03e/pytest.log
ORBX_FAST_X2=1 timeout 300 python3 -m pytest tests/test_gpu_extractor.py -m gpu -x -q -k "euroc or kitti or tumvi or flat_and_noise or parameter_sweep or stagewise" > gpurun_out/r03e/pytest_x2.log 2>&1
echo "pytest x2 rc=$?"; tail -3 gpurun_out/r03e/pytest_x2.log
bash tools/ab.sh "ORBX_PYR_CHAIN=0" "ORBX_NONE=1" "ORBX_FAST_X2=1" 2>&1 | tee gpurun_out/r03e/ab.log
for v in "ORBX_NONE=1" "ORBX_FAST_X2=1" "ORBX_PYR_CHAIN=0"; do
  r=$(env $v python3 bench.py --workload tumvi --steps 30 --warmup 5 --cpu-frames 0 --no-profile --verify 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "tumvi $v : $r" | tee -a gpurun_out/r03e/ab.log
done
ORBX_FAST_X2=1 bash tools/quick_prof.sh euroc sq > /dev/null 2>&1; cp gpurun_out/qp/stats.csv gpurun_out/r03e/stats_x2_chain.csv; cp gpurun_out/qp/qp_pmc_sq_counters.csv gpurun_out/r03e/sq_x2_chain.csv
head -16 gpurun_out/r03e/stats_x2_chain.csv; grep "strip\|march\|list" gpurun_out/r03e/sq_x2_chain.csv
python3 tools/rocprof_dispatches.py gpurun_out/qp/se/se_results.db | grep -i "march\|strip" | tee gpurun_out/r03e/dispatches.txt
