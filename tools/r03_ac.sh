#!/bin/bash
# visit ac: the round's new stereo code in guard mode (every device buffer against an unmapped range, poison-filled) + batch-size sweep for the record
mkdir -p gpurun_out/r03ac
run() { wl=$1; shift; timeout 300 python bench.py --workload $wl --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']; print('$wl $*', d['value'], d['ms_per_step'], r['median'], r['min'], r['max'], d.get('parity_checked'))"; }
{
for g in 1 2; do
  for fill in 0xA5 0x00; do
    ORBX_GUARD=$g ORBX_GUARD_FILL=$fill timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_matcher.py tests/test_gpu_adapter_vs_reference.py -x -q -m gpu -k "stereo" 2>&1 | tail -1
  done
  ORBX_GUARD=$g timeout 300 python bench.py --workload kitti --steps 6 --warmup 2 --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 2>&1 | tail -1 | cut -c1-200
done
run kitti --batch 128
run tumvi --batch 128
run euroc --batch 512
} > gpurun_out/r03ac/log.txt 2>&1
cat gpurun_out/r03ac/log.txt
