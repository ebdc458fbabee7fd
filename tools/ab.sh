#!/bin/bash
# Hardware A/B of environment switches on the bench's timed loop (one script instead of one per experiment).
#   usage: bash tools/ab.sh [-t tag] [-w "euroc kitti tumvi"] [-r reps] [-p] [-v] VARIANT...
#     VARIANT   environment assignments in one word list, e.g. "ORBX_FUSED_BLUR=0" or "ORBX_NONE=1" (the default build) or "A=1 B=2"
#     -r reps   interleaved repetitions of the whole variant list (default 2): box drift shows up as a difference between repetitions
#     -p        also a serialized rocprofv3 --kernel-trace --stats pass per variant (euroc child) -> <tag>_<i>_kernel_stats.csv
#     -v        keep the bench's full parity check on (default: --verify 4 stays on; -V turns it off for speed)
#   output: gpurun_out/<tag>/ab.log (one line per run: workload, variant, value, ms/step first region, median / min / max over the regions, parity)
set -u
TAG=ab; WLS="euroc"; REPS=2; PROF=0; VERIFY=4
while getopts "t:w:r:pvV" o; do case $o in t) TAG=$OPTARG;; w) WLS=$OPTARG;; r) REPS=$OPTARG;; p) PROF=1;; v) VERIFY=4;; V) VERIFY=0;; *) exit 2;; esac; done
shift $((OPTIND - 1))
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
run() { wl=$1; shift
  env "$@" timeout 300 python3 bench.py --workload $wl --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 --no-profile --verify $VERIFY 2>$O/err.txt | tail -1 | \
    python3 -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['repeats']['ms_per_step']; p=d.get('pcie_inclusive') or {}
    print('$wl | $* |', d['value'], '| ms', d['ms_per_step'], 'median', r['median'], 'min', r['min'], 'max', r['max'], '| host-input ms', p.get('ms_per_step'), '| parity', bool(d.get('parity_checked')))
except Exception as e:
    print('$wl | $* | FAILED', e)"
  grep -h "PARITY\|rror" $O/err.txt | head -3; }
{
for rep in $(seq $REPS); do for wl in $WLS; do for v in "$@"; do run $wl $v; done; done; done
if [ $PROF = 1 ]; then i=0; for v in "$@"; do i=$((i+1))
  env $v timeout 90 rocprofv3 --kernel-trace --stats -d $O/se$i -o se -- python3 bench.py --pmc-child --workload euroc --steps 12 --warmup 3 > /dev/null 2>&1
  db=$(find $O/se$i -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
  echo "--- serialized kernel stats: $v"; python3 tools/rocprof_summary.py $db $O/${TAG}_${i}_kernel_stats.csv | head -24; rm -rf $O/se$i
done; fi
} 2>&1 | tee $O/ab.log
