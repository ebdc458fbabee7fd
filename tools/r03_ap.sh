#!/bin/bash
# visit ap: k_finalize + k_describe in one launch (k_finalize_describe_ordered) against two launches
mkdir -p gpurun_out/r03ap
run() { wl=$1; shift; env "$@" timeout 200 python bench.py --workload $wl --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']; print('$wl $*', d['value'], d['ms_per_step'], r['median'], r['min'], r['max'], d.get('parity_checked'))"; }
{
timeout 300 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -1
run euroc ORBX_FIN_DESC=0; run euroc ORBX_NONE=1; run euroc ORBX_FIN_DESC=0; run euroc ORBX_NONE=1
run kitti ORBX_FIN_DESC=0; run kitti ORBX_NONE=1; run tumvi ORBX_FIN_DESC=0; run tumvi ORBX_NONE=1
} > gpurun_out/r03ap/log.txt 2>&1
cat gpurun_out/r03ap/log.txt
