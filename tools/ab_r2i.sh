#!/bin/bash
# visit 4: two-column resize kernel: extractor / pipeline tests, then A/B against the one-column kernel, serialized kernel durations
mkdir -p gpurun_out/r2i
export TMPDIR=/tmp
timeout 200 python3 -m pytest tests/test_gpu_extractor.py tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/r2i/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2i/pytest.log
bash tools/ab.sh "ORBX_NONE=1" "ORBX_RESIZE_COLS=1" 2>&1 | tee gpurun_out/r2i/ab.log
bash tools/quick_prof.sh > /dev/null 2>&1; cat gpurun_out/qp/stats.csv | head -8 | tee gpurun_out/r2i/prof.log
