#!/bin/bash
# serialized kernel stats of the bench child for each variant (no parity): usage bash tools/r05_prof.sh <tag> VARIANT...
TAG=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
i=0; for v in "$@"; do i=$((i+1))
  env $v timeout 90 rocprofv3 --kernel-trace --stats -d $O/se$i -o se -- python3 bench.py --pmc-child --workload ${WL:-euroc} --steps 8 --warmup 2 > /dev/null 2>&1
  db=$(find $O/se$i -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
  echo "--- $v"; python3 tools/rocprof_summary.py $db $O/${TAG}_${i}_kernel_stats.csv | grep -i "${GREP:-k_}" | head -${LINES_:-4}; rm -rf $O/se$i
done 2>&1 | tee $O/prof.log
