#!/bin/bash
# A/B of environment switches on the TUM-VI workload (extract + SearchByProjection vs 10 000 map points)
for v in "$@"; do
  for rep in 1 2 3; do
    r=$(env $v python3 bench.py --workload tumvi --steps 30 --warmup 5 --cpu-frames 0 --no-profile --verify 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'))")
    echo "$v : $r"
  done
done
