#!/bin/bash
# GPU visit: k_fast_strip on the hardware -- parity (extractor + pipeline GPU tests), A/B against the per-cell k_fast_ini, polarity split on / off,
# queue capacities (LDS -> workgroups per CU), serialized kernel profile + SQ counters of the default
mkdir -p gpurun_out/r03b
export TMPDIR=/tmp
timeout 400 python3 -m pytest tests/test_gpu_extractor.py tests/test_gpu_pipeline.py -m gpu -x -q > gpurun_out/r03b/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r03b/pytest.log
bash tools/ab.sh "ORBX_FAST_STRIP=0" "ORBX_NONE=1" "ORBX_FAST_POL=0" "ORBX_STRIP_GCAP=320 ORBX_STRIP_QCAP=640" "ORBX_STRIP_GCAP=768 ORBX_STRIP_QCAP=1024" 2>&1 | tee gpurun_out/r03b/ab.log
bash tools/quick_prof.sh euroc sq > /dev/null 2>&1; cp gpurun_out/qp/stats.csv gpurun_out/r03b/stats.csv; cp gpurun_out/qp/qp_pmc_sq_counters.csv gpurun_out/r03b/ 2>/dev/null
head -16 gpurun_out/r03b/stats.csv; cat gpurun_out/r03b/qp_pmc_sq_counters.csv 2>/dev/null | head -20
