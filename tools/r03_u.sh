#!/bin/bash
# visit u: where the KITTI and TUM-VI steps go (timeline of one step each)
mkdir -p gpurun_out/r03u
{ bash tools/workload_timeline.sh kitti; bash tools/workload_timeline.sh tumvi; } > gpurun_out/r03u/log.txt 2>&1
cat gpurun_out/r03u/log.txt | head -250
