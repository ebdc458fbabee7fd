#!/bin/bash
mkdir -p gpurun_out/r03n
for n in 0 1 2 3; do
  ORBX_SPARE_STREAMS=$n bash tools/ab.sh "ORBX_SPARE_STREAMS=$n" 2>&1 | head -1 | tee -a gpurun_out/r03n/ab.log
  for wl in kitti tumvi tumvi; do
    r=$(ORBX_SPARE_STREAMS=$n python3 bench.py --workload $wl --steps 30 --warmup 5 --cpu-frames 0 --no-profile --verify 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "$wl spare=$n : $r" | tee -a gpurun_out/r03n/ab.log
  done
done
