#!/bin/bash
# ms per step of the timed region as a function of its length K (the contract's K = 20 against the steady state): constant start / drain cost or a clock ramp?
cd "$(dirname "$0")/.."
for k in 20 50 100 300 1000; do
  python3 bench.py --steps $k --warmup 5 --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 --no-profile --repeat 3 --verify 0 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); g=d['repeats']['step_gaps'][0]; print('K = $k: ms_per_step', d['ms_per_step'], d['repeats']['ms_per_step_in_order'], 'region total ms', round(d['ms_per_step']*$k,2), 'median gap', g['median_gap_ms'], 'max gap', g['max_gap_ms'], 'at', g['argmax'])"
done
