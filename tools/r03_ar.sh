#!/bin/bash
# visit ar: the stereo test on an image taller than the row index (row_shift = 1) on the hardware
mkdir -p gpurun_out/r03ar
timeout 120 python -m pytest tests/test_gpu_matcher.py -x -q -m gpu -k "taller or stereo" 2>&1 | tail -2 > gpurun_out/r03ar/log.txt
cat gpurun_out/r03ar/log.txt
