#!/bin/bash
# GPU visit: k_pyr_resize_march (rows in flight, rows per block), k_fast_strip with 1 / 2 / 4 waves per strip, runtime priming at configuration
mkdir -p gpurun_out/r03c
export TMPDIR=/tmp
timeout 300 python3 -m pytest tests/test_gpu_extractor.py -m gpu -x -q > gpurun_out/r03c/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r03c/pytest.log
bash tools/ab.sh "ORBX_FAST_STRIP=0 ORBX_RESIZE_MARCH=0" "ORBX_FAST_STRIP=0" "ORBX_FAST_STRIP=0 ORBX_RESIZE_MARCH=4" "ORBX_FAST_STRIP=0 ORBX_RESIZE_RB=16" "ORBX_FAST_STRIP=0 ORBX_RESIZE_RB=64" \
   "ORBX_NONE=1" "ORBX_STRIP_WAVES=2 ORBX_STRIP_GCAP=1024 ORBX_STRIP_QCAP=1632" "ORBX_STRIP_WAVES=1 ORBX_STRIP_GCAP=2048 ORBX_STRIP_QCAP=3264" "ORBX_FAST_STRIP=0 ORBX_PRIME=12" 2>&1 | tee gpurun_out/r03c/ab.log
ORBX_FAST_STRIP=0 bash tools/quick_prof.sh euroc > /dev/null 2>&1; cp gpurun_out/qp/stats.csv gpurun_out/r03c/stats_march.csv; head -16 gpurun_out/r03c/stats_march.csv
