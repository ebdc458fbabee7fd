#!/bin/bash
# Round measurement on the GPU box: bench line (with in-run PMC traffic), rocprofv3 kernel stats (overlapped + serialized),
# PMC traffic and SQ counters of every kernel.
# usage: bash tools/measure_round.sh <tag>      (writes profiles/<tag>_*; run through gpurun, then commit profiles/)
set -u
TAG=${1:-r02_x}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/measure_$TAG
rm -rf $O; mkdir -p $O profiles
biggest_db() { find "$1" -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2; }
CHILD="python bench.py --pmc-child --workload euroc"
# overlapped pipeline exactly as timed (launcher -> rank process; the rank's database is the big one)
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ov -o ov -- python bench.py --steps 12 --warmup 3 --cpu-frames 0 --no-profile --verify 0 > /dev/null 2>&1
python tools/rocprof_summary.py $(biggest_db $O/ov) profiles/${TAG}_overlapped_kernel_stats.csv > /dev/null
# serialized (every kernel on the extractor's main stream): the per-kernel durations the roofline is computed from
ORBX_SIDE_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/se -o se -- $CHILD --steps 12 --warmup 3 > /dev/null 2>&1
python tools/rocprof_summary.py $(biggest_db $O/se) profiles/${TAG}_serialized_kernel_stats.csv
ORBX_SIDE_STREAMS=0 timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/pf -o pf -- $CHILD --steps 3 --warmup 1 > /dev/null 2>&1
ORBX_SIDE_STREAMS=0 timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pw -o pw -- $CHILD --steps 3 --warmup 1 > /dev/null 2>&1
python tools/pmc_summary.py $(biggest_db $O/pf) $(biggest_db $O/pw) $TAG
ORBX_SIDE_STREAMS=0 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/sq -o sq -- $CHILD --steps 3 --warmup 1 > /dev/null 2>&1
python tools/pmc_sq.py $TAG $(biggest_db $O/sq)
# the bench line, with roofline.traffic measured by its own rocprofv3 --pmc child passes
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err | tail -1 > profiles/${TAG}_bench_euroc.json
cut -c1-600 profiles/${TAG}_bench_euroc.json
cp -r profiles $O/profiles_copy
