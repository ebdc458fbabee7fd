#!/bin/bash
mkdir -p gpurun_out/r03f
export TMPDIR=/tmp
timeout 300 python3 -m pytest tests/test_gpu_extractor.py -m gpu -x -q > gpurun_out/r03f/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r03f/pytest.log
bash tools/ab.sh "ORBX_NONE=1" "ORBX_MATCH_PRIO=high" 2>&1 | tee gpurun_out/r03f/ab.log
bash tools/timeline.sh > gpurun_out/r03f/timeline.txt 2>&1; head -70 gpurun_out/r03f/timeline.txt
bash tools/quick_prof.sh euroc > /dev/null 2>&1; cp gpurun_out/qp/stats.csv gpurun_out/r03f/stats.csv; head -16 gpurun_out/r03f/stats.csv
python3 tools/rocprof_dispatches.py gpurun_out/qp/se/se_results.db | grep -i "march\|strip" | tee gpurun_out/r03f/dispatches.txt
