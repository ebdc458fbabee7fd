#!/bin/bash
# end-of-round evidence, most important first (the GPU budget of the round is nearly spent): bench lines, serialized kernel stats,
# PMC traffic, then the overlapped stats and SQ counters.  usage: bash tools/final_r02f.sh <tag>
set -u
TAG=${1:-r02_f}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/measure_$TAG
rm -rf $O; mkdir -p $O/profiles_copy profiles
biggest_db() { find "$1" -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2; }
CHILD="python bench.py --pmc-child --workload euroc"
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err | tail -1 > $O/profiles_copy/${TAG}_bench_euroc.json
cut -c1-400 $O/profiles_copy/${TAG}_bench_euroc.json; echo
ORBX_SIDE_STREAMS=0 timeout 60 rocprofv3 --kernel-trace --stats -d $O/se -o se -- $CHILD --steps 12 --warmup 3 > /dev/null 2>&1
python tools/rocprof_summary.py $(biggest_db $O/se) $O/profiles_copy/${TAG}_serialized_kernel_stats.csv
for wl in tumvi kitti; do
  timeout 120 python bench.py --workload $wl 2> $O/bench_$wl.err | tail -1 > $O/profiles_copy/${TAG}_bench_$wl.json
  cut -c1-260 $O/profiles_copy/${TAG}_bench_$wl.json; echo
done
ORBX_SIDE_STREAMS=0 timeout 60 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/pf -o pf -- $CHILD --steps 3 --warmup 1 > /dev/null 2>&1
ORBX_SIDE_STREAMS=0 timeout 60 rocprofv3 --pmc WRITE_SIZE -d $O/pw -o pw -- $CHILD --steps 3 --warmup 1 > /dev/null 2>&1
python tools/pmc_summary.py $(biggest_db $O/pf) $(biggest_db $O/pw) $TAG && cp profiles/${TAG}_pmc_traffic.csv $O/profiles_copy/
ORBX_SIDE_STREAMS=0 timeout 60 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/sq -o sq -- $CHILD --steps 3 --warmup 1 > /dev/null 2>&1
python tools/pmc_sq.py $TAG $(biggest_db $O/sq) > /dev/null && cp profiles/${TAG}_pmc_sq_counters.csv $O/profiles_copy/
timeout 60 rocprofv3 --kernel-trace --stats -d $O/ov -o ov -- python bench.py --steps 12 --warmup 3 --cpu-frames 0 --no-profile --verify 0 --no-pmc > /dev/null 2>&1
python tools/rocprof_summary.py $(biggest_db $O/ov) $O/profiles_copy/${TAG}_overlapped_kernel_stats.csv > /dev/null
rm -rf $O/se $O/pf $O/pw $O/sq $O/ov
ls $O/profiles_copy
