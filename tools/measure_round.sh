#!/bin/bash
# Round measurement on the GPU box: bench line, rocprofv3 kernel stats (overlapped + serialized), PMC traffic and SQ counters.
# usage: bash tools/measure_round.sh <tag>      (writes profiles/<tag>_*; run through gpurun, then commit profiles/)
set -u
TAG=${1:-r01_x}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/measure_$TAG
rm -rf $O; mkdir -p $O profiles
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ov -o ov -- python bench.py --steps 12 --warmup 3 --cpu-frames 0 --no-profile > /dev/null 2>&1
python tools/rocprof_summary.py $(find $O/ov -name "*.db" | head -1) profiles/${TAG}_overlapped_kernel_stats.csv > /dev/null
ORBX_SIDE_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/se -o se -- python bench.py --steps 12 --warmup 3 --cpu-frames 0 --no-profile > /dev/null 2>&1
python tools/rocprof_summary.py $(find $O/se -name "*.db" | head -1) profiles/${TAG}_serialized_kernel_stats.csv
ORBX_SIDE_STREAMS=0 timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pf -o pf -- python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-profile > /dev/null 2>&1
ORBX_SIDE_STREAMS=0 timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pw -o pw -- python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-profile > /dev/null 2>&1
python tools/pmc_summary.py $(find $O/pf -name "*.db" | head -1) $(find $O/pw -name "*.db" | head -1) $TAG
# the bench line after the PMC passes: its roofline.traffic is read from the pmc_traffic.json they just wrote (same batch size)
timeout 600 python bench.py 2> $O/bench.err | tail -1 > profiles/${TAG}_overlapped_bench.json
cat profiles/${TAG}_overlapped_bench.json | cut -c1-400
ORBX_SIDE_STREAMS=0 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/sq -o sq -- python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-profile > /dev/null 2>&1
python tools/pmc_sq.py $TAG $(find $O/sq -name "*.db" | head -1)
cp -r profiles $O/profiles_copy
