#!/bin/bash
# The open 1007-vs-1008 item: failure RATE of the short-batch pipeline (tests/test_gpu_pipeline.py as a script:
# 16 frames per batch, 4 steps, 1024^2 canvas) under the settings that tell an ordering problem from a data-dependent one.
# usage: bash tools/open_item_probe.sh [runs per setting, default 6]
N=${1:-6}
IFS="|" read -ra SETTINGS <<< "${ORBX_PROBE_SETTINGS:-ORBX_NONE=1|ORBX_ENSURE_NOSYNC=1|ORBX_SIDE_STREAMS=0|AMD_SERIALIZE_KERNEL=3|HIP_LAUNCH_BLOCKING=1|ORBX_OCTREE=seq|ORBX_COPY_AFTER_MATCH=1|GPU_MAX_HW_QUEUES=8}"
for v in "${SETTINGS[@]}"; do
  ok=0; bad=0; first=""
  for i in $(seq 1 $N); do
    out=$(env $v timeout 120 python3 tests/test_gpu_pipeline.py 16 4 small 2>&1 | tail -3)
    if echo "$out" | grep -q "pipeline ok"; then ok=$((ok+1)); else bad=$((bad+1)); [ -z "$first" ] && first=$(echo "$out" | tr '\n' ' ' | cut -c1-200); fi
  done
  echo "$v : ok $ok failed $bad   $first"
done
