#!/bin/bash
# round-2 session-2 visit: GPU test-suite with the new kernels, fallback attribution on failure, A/B of the new switches, kernel durations
mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
timeout 300 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2f/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -12 gpurun_out/r2f/pytest.log
if [ $rc -ne 0 ]; then
  for v in ORBX_BLUR_KERNEL=0 ORBX_PYR_XCD=0 ORBX_RESIZE_PK=0 ORBX_RESOLVE_RESCAN=full; do
    env $v timeout 200 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2f/pytest_$v.log 2>&1
    echo "pytest with $v rc=$?"; tail -3 gpurun_out/r2f/pytest_$v.log
  done
fi
bash tools/ab.sh "ORBX_NONE=1" "ORBX_BLUR_KERNEL=0" "ORBX_PYR_XCD=0" "ORBX_RESIZE_PK=0" "ORBX_RESOLVE_RESCAN=full" "ORBX_BLUR_SIDE=0" 2>&1 | tee gpurun_out/r2f/ab.log
bash tools/quick_prof.sh 2>&1 | tee gpurun_out/r2f/prof.log; cat gpurun_out/qp/stats.csv | tee -a gpurun_out/r2f/prof.log
