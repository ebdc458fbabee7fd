#!/bin/bash
# serialized kernel stats of the texture stress scene (tuned queues)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/${1:-texprof}; mkdir -p $O
timeout 120 rocprofv3 --kernel-trace --stats -d $O/se -o se -- python3 bench.py --pmc-child --workload euroc --scene texture --steps 6 --warmup 2 > /dev/null 2>&1
db=$(find $O/se -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
python3 tools/rocprof_summary.py $db $O/texture_kernel_stats.csv | head -16 | tee $O/texprof.log; rm -rf $O/se
