#!/bin/bash
mkdir -p gpurun_out/r03h
export TMPDIR=/tmp
export ORBX_COPY_BLOCKS=0
bash tools/ab.sh "ORBX_NONE=1" "GPU_FORCE_BLIT_COPY_SIZE=0" "GPU_FORCE_BLIT_COPY_SIZE=0 HSA_ENABLE_SDMA_GANG=0" "HSA_ENABLE_SDMA=0" 2>&1 | tee gpurun_out/r03h/ab.log
GPU_FORCE_BLIT_COPY_SIZE=0 TL_COPIES=1 bash tools/timeline.sh > gpurun_out/r03h/timeline_sdma.txt 2>&1; head -60 gpurun_out/r03h/timeline_sdma.txt
