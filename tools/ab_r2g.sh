#!/bin/bash
# round-2 session-2, visit 2: pyramid-ahead pipeline + copy order; GPU test-suite, A/B of the switches, the three workloads
mkdir -p gpurun_out/r2g
export TMPDIR=/tmp
timeout 300 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2g/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -12 gpurun_out/r2g/pytest.log
if [ $rc -ne 0 ]; then
  for v in ORBX_PYR_AHEAD=0 ORBX_PYR_AHEAD=1 ORBX_COPY_AFTER_MATCH=1; do
    env $v timeout 200 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2g/pytest_$v.log 2>&1
    echo "pytest with $v rc=$?"; tail -3 gpurun_out/r2g/pytest_$v.log
  done
fi
bash tools/ab.sh "ORBX_NONE=1" "ORBX_PYR_AHEAD=1" "ORBX_PYR_AHEAD=0" "ORBX_COPY_AFTER_MATCH=1" "ORBX_PYR_AHEAD=0 ORBX_COPY_AFTER_MATCH=1" "ORBX_STREAM_PRIO=1" "ORBX_STREAM_PRIO=2" "ORBX_BLUR_SIDE=0" 2>&1 | tee gpurun_out/r2g/ab.log
bash tools/ab_wl.sh 2>&1 | tee gpurun_out/r2g/wl.log
