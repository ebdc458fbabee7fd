#!/bin/bash
# end-of-round evidence: full measurement of the euroc workload + the kitti / tumvi bench lines (with cpu_baseline)
TAG=${1:-r02_b}
bash tools/measure_round.sh $TAG
O=gpurun_out/measure_$TAG
for wl in kitti tumvi; do
  timeout 600 python bench.py --workload $wl 2> $O/bench_$wl.err | tail -1 > $O/profiles_copy/${TAG}_bench_$wl.json
  cut -c1-300 $O/profiles_copy/${TAG}_bench_$wl.json
done
