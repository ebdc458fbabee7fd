#!/bin/bash
# visit r: where the 0.25 ms of a single-frame call go
mkdir -p gpurun_out/r03r
{
for e in ORBX_NONE=1 ORBX_SIDE_STREAMS=0; do
  for i in 1 2; do echo "$e : $(env $e python tools/latency.py)"; done
done
bash tools/latency_timeline.sh ORBX_NONE=1
bash tools/latency_timeline.sh ORBX_SIDE_STREAMS=0
} > gpurun_out/r03r/log.txt 2>&1
cat gpurun_out/r03r/log.txt
