#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06_${1:-d}; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_extractor.py tests/test_gpu_pipeline.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
bash tools/density_sweep.sh ${1:-d} 2>&1 | cut -c1-600
