#!/bin/bash
# round-5 GPU visit: selected GPU tests, A/B of the pyramid variants on the bench's timed loop, serialized kernel stats of the first variants
# usage: bash tools/r05_visit.sh <tag> <n profiled variants> "<pytest args>" VARIANT...
set -u
TAG=$1; NPROF=$2; PYT=$3; shift 3
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
if [ -n "$PYT" ]; then timeout 900 python3 -m pytest $PYT -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log; fi
run() { env "$@" timeout 200 python3 bench.py --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 --no-profile --verify ${VERIFY:-4} --repeat 3 2>$O/err.txt | tail -1 | \
    python3 -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']; p=d.get('pcie_inclusive') or {}
    print('$* |', d['value'], '| ms', d['ms_per_step'], 'median', r['median'], 'min', r['min'], 'max', r['max'], '| host-input ms', p.get('ms_per_step'), '| parity', bool(d.get('parity_checked')))
except Exception as e:
    print('$* | FAILED', e)"
  grep -h "PARITY\|rror" $O/err.txt | head -3; }
{
for v in "$@"; do run $v; done
i=0; for v in "$@"; do i=$((i+1)); [ $i -gt $NPROF ] && break
  env $v timeout 90 rocprofv3 --kernel-trace --stats -d $O/se$i -o se -- python3 bench.py --pmc-child --workload euroc --steps 12 --warmup 3 > /dev/null 2>&1
  db=$(find $O/se$i -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
  echo "--- serialized kernel stats: $v"; python3 tools/rocprof_summary.py $db $O/${TAG}_${i}_kernel_stats.csv | head -22; rm -rf $O/se$i
done
} 2>&1 | tee $O/ab.log
