#!/bin/bash
for b in 256 384 512 768; do
  r=$(python3 bench.py --steps 20 --warmup 4 --cpu-frames 0 --no-profile --verify 0 --batch $b 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['pcie_inclusive']['value'])")
  echo "batch $b : $r"
done
