#!/bin/bash
# round 4, GPU visit 5: the TUM-VI slow regions -- per-region order, largest gap between step completions, clocks / power sampled beside a run
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04v5; mkdir -p $O
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['repeats']
print('$1', d['value'], 'order', r['ms_per_step_in_order']); [print('   ', g) for g in r['step_gaps']]"; }
for i in 1 2 3 4; do
  timeout 300 python3 bench.py --workload tumvi --cpu-frames 0 --no-pmc --no-profile --repeat 10 > $O/tumvi_$i.json 2> $O/tumvi_$i.err; show $O/tumvi_$i.json
done
( while true; do rocm-smi --showclocks --showpower --showtemp --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' '; echo; sleep 0.1; done ) > $O/smi.log 2>&1 &
SMI=$!
timeout 300 python3 bench.py --workload tumvi --cpu-frames 0 --no-pmc --no-profile --repeat 12 > $O/tumvi_smi.json 2> $O/tumvi_smi.err; show $O/tumvi_smi.json
kill $SMI
wc -l $O/smi.log; head -3 $O/smi.log; python3 - <<'PY'
import re
rows=[l for l in open('gpurun_out/r04v5/smi.log') if l.strip()]
print(rows[0][:300])
import collections
vals=collections.Counter()
for l in rows:
    m=re.findall(r'\((\d+)Mhz\)', l)
    if m: vals[tuple(m[:2])]+=1
print(vals.most_common(12))
PY
for i in 1 2; do timeout 300 python3 bench.py --workload euroc --cpu-frames 0 --no-pmc --no-profile --no-other-workloads --latency 0 --repeat 10 > $O/euroc_$i.json 2> $O/euroc_$i.err; show $O/euroc_$i.json; done
