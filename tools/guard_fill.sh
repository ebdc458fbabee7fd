#!/bin/bash
# guard mode with different poison bytes: uninitialised counts / indices that happen to be benign for one fill are not for another
mkdir -p gpurun_out/guard
for fill in 127 1 64 255 3; do
  ORBX_GUARD=1 ORBX_GUARD_FILL=$fill timeout 300 python3 bench.py --gpus 1 --steps 4 --warmup 1 --cpu-frames 0 > gpurun_out/guard/bench_f$fill.out 2> gpurun_out/guard/bench_f$fill.err
  echo "bench fill=$fill rc=$?" | tee -a gpurun_out/guard/summary_fill.txt
  tail -c 300 gpurun_out/guard/bench_f$fill.err
done
for fill in 127 1 3; do
  ORBX_GUARD=1 ORBX_GUARD_FILL=$fill timeout 900 python3 -m pytest tests -m gpu -q > gpurun_out/guard/pytest_f$fill.out 2>&1
  echo "pytest fill=$fill rc=$?" | tee -a gpurun_out/guard/summary_fill.txt
  tail -8 gpurun_out/guard/pytest_f$fill.out
done
