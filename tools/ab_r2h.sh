#!/bin/bash
# visit 3: hardware-queue count (streams beyond GPU_MAX_HW_QUEUES share a queue and serialize) x pyramid-ahead; then the GPU test-suite
mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
bash tools/ab.sh "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=8 ORBX_PYR_AHEAD=1" "GPU_MAX_HW_QUEUES=8 ORBX_PYR_AHEAD=0" "ORBX_NONE=1" "GPU_MAX_HW_QUEUES=8 ORBX_BLUR_SIDE=0" "GPU_MAX_HW_QUEUES=16" 2>&1 | tee gpurun_out/r2h/ab.log
timeout 300 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2h/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -8 gpurun_out/r2h/pytest.log
