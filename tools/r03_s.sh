#!/bin/bash
# visit s: single-frame call with the host mirror (one synchronisation), row blocks of 8 for small batches, zero-copy input
mkdir -p gpurun_out/r03s
{
timeout 600 python -m pytest tests/test_gpu_extractor.py -x -q -m gpu 2>&1 | tail -3
for e in ORBX_NONE=1 ORBX_RB_SMALL=0 ORBX_RB_SMALL=16 ORBX_RB_SMALL=4 ORBX_ZC_IN=1 ; do
  for i in 1 2; do echo "$e : $(env $e python tools/latency.py 2>/dev/null)"; done
done
bash tools/latency_timeline.sh ORBX_NONE=1
bash tools/latency_timeline.sh ORBX_ZC_IN=1
} > gpurun_out/r03s/log.txt 2>&1
cat gpurun_out/r03s/log.txt
