#!/bin/bash
# visit aj: the driver's command after the time-based settle (first timed region of the embedded KITTI / TUM-VI runs)
mkdir -p gpurun_out/r03aj
{
for i in 1 2 3; do
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r03aj/bench$i.err | tail -1 > gpurun_out/r03aj/bench$i.json ) 2>&1 | grep real
python3 - $i <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r03aj/bench{sys.argv[1]}.json').read())
print('euroc', d['value'], d['ms_per_step'], d['settle_steps'], d['repeats']['ms_per_step'])
for k,v in d['other_workloads'].items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('settle_steps'), v.get('repeats',{}).get('ms_per_step'))
PY
done
} > gpurun_out/r03aj/log.txt 2>&1
cat gpurun_out/r03aj/log.txt
