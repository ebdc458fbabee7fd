#!/bin/bash
for wl in euroc tumvi kitti; do
 for rep in 1 2 3; do
  r=$(python3 bench.py --workload $wl --steps 30 --warmup 5 --cpu-frames 0 --no-profile --verify 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$wl : $r"
 done
done
