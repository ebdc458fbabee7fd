#!/bin/bash
# visit ak: the new chain test (odd batch sizes, two geometries)
mkdir -p gpurun_out/r03ak
timeout 600 python -m pytest tests/test_gpu_extractor.py -x -q -m gpu -k "chain_odd" 2>&1 | tail -5 > gpurun_out/r03ak/log.txt
cat gpurun_out/r03ak/log.txt
