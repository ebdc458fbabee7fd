#!/bin/bash
# A/B of environment switches on the timed bench loop (no profile / cpu / verify): prints value + ms_per_step per variant
for v in "$@"; do
  for rep in 1 2; do
    r=$(env $v python3 bench.py --steps 30 --warmup 5 --cpu-frames 0 --no-profile --verify 0 --repeat 1 --latency 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['pcie_inclusive']['ms_per_step'], 'host_enqueue', d.get('host_enqueue_ms_per_step'), 'settle', d.get('settle_ms_per_step'))")
    echo "$v : $r"
  done
done
