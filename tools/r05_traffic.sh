#!/bin/bash
# serialized kernel stats + PMC traffic of the bench child: usage bash tools/r05_traffic.sh <tag> [workload]
TAG=$1; WL=${2:-euroc}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
biggest_db() { find "$1" -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2; }
CHILD="python bench.py --pmc-child --workload $WL"
timeout 60 rocprofv3 --kernel-trace --stats -d $O/se -o se -- $CHILD --steps 12 --warmup 3 > /dev/null 2>&1
python tools/rocprof_summary.py $(biggest_db $O/se) $O/${TAG}_serialized_kernel_stats.csv | head -16
timeout 60 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/pf -o pf -- $CHILD --steps 3 --warmup 1 > /dev/null 2>&1
timeout 60 rocprofv3 --pmc WRITE_SIZE -d $O/pw -o pw -- $CHILD --steps 3 --warmup 1 > /dev/null 2>&1
python tools/pmc_summary.py $(biggest_db $O/pf) $(biggest_db $O/pw) $TAG && mv profiles/${TAG}_pmc_traffic.csv $O/ && cat $O/${TAG}_pmc_traffic.csv
rm -rf $O/se $O/pf $O/pw
