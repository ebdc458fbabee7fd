#!/bin/bash
# A/B of two builds of the library (orb_slam3_amd/liborbx_base.so, liborbx_new.so) on the timed loop + serialized kernel stats
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
run() { python3 bench.py --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 --no-profile --verify 4 --workload $1 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']; print('$1 $2', d['value'], 'ms', d['ms_per_step'], 'median', r['median'], 'min', r['min'], 'max', r['max'], bool(d['parity_checked']))"; }
for rep in 1 2; do for v in base new; do cp orb_slam3_amd/liborbx_$v.so orb_slam3_amd/liborbx.so; for wl in ${WLS:-euroc}; do run $wl $v; done; done; done
for v in base new; do cp orb_slam3_amd/liborbx_$v.so orb_slam3_amd/liborbx.so
  timeout 90 rocprofv3 --kernel-trace --stats -d gpurun_out/abl/se$v -o se -- python3 bench.py --pmc-child --workload euroc --steps 12 --warmup 3 > /dev/null 2>&1
  db=$(find gpurun_out/abl/se$v -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
  echo "--- serialized kernel stats: $v"; python3 tools/rocprof_summary.py $db gpurun_out/abl/${v}_kernel_stats.csv | head -8; rm -rf gpurun_out/abl/se$v
done
cp orb_slam3_amd/liborbx_new.so orb_slam3_amd/liborbx.so
