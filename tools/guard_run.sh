#!/bin/bash
# Guard-mode runs of the bench and of the GPU tests: every liborbx device buffer ends (ORBX_GUARD=1) or starts (ORBX_GUARD=2)
# at an unmapped address range and is poison-filled, so over-/under-runs and uninitialised indices fault deterministically.
mkdir -p gpurun_out/guard
for g in 1 2; do
  ORBX_GUARD=$g timeout 300 python3 bench.py --gpus 1 --steps 6 --warmup 2 --cpu-frames 0 > gpurun_out/guard/bench_g$g.out 2> gpurun_out/guard/bench_g$g.err
  echo "bench guard=$g rc=$?" | tee -a gpurun_out/guard/summary.txt
  tail -c 600 gpurun_out/guard/bench_g$g.err
done
for g in 1 2; do
  ORBX_GUARD=$g timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/guard/pytest_g$g.out 2>&1
  echo "pytest guard=$g rc=$?" | tee -a gpurun_out/guard/summary.txt
  tail -5 gpurun_out/guard/pytest_g$g.out
done
