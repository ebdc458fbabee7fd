#!/bin/bash
# visit ah: pyramid chain + FAST strips of level 0 in ONE launch (k_pyr_chain_fast0_ordered) against the chain alone
export TMPDIR=/tmp
mkdir -p gpurun_out/r03ah
run() { wl=$1; shift; env "$@" timeout 200 python bench.py --workload $wl --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 $EXTRA 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']; print('$wl $* $EXTRA', d['value'], d['ms_per_step'], r['median'], r['min'], r['max'], d.get('parity_checked'))"; }
{
timeout 300 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_extractor.py -x -q -m gpu 2>&1 | tail -2
EXTRA=""
run euroc ORBX_CHAIN_FAST0=0; run euroc ORBX_NONE=1; run euroc ORBX_CHAIN_FAST0=0; run euroc ORBX_NONE=1
EXTRA="--batch 128"
for wl in kitti tumvi; do run $wl ORBX_CHAIN_FAST0=0; run $wl ORBX_NONE=1; done
for v in 0 1; do
  ORBX_CHAIN_FAST0=$v ORBX_SIDE_STREAMS=0 timeout 90 rocprofv3 --kernel-trace --stats -d gpurun_out/r03ah/se$v -o se -- python bench.py --pmc-child --workload euroc --steps 12 --warmup 3 > /dev/null 2>&1
  python tools/rocprof_summary.py $(find gpurun_out/r03ah/se$v -name "*.db" | head -1) gpurun_out/r03ah/serialized_$v.csv | grep -i "pyr\|fast"
  rm -rf gpurun_out/r03ah/se$v
done
bash tools/timeline.sh | head -26
} > gpurun_out/r03ah/log.txt 2>&1
cat gpurun_out/r03ah/log.txt
