#!/bin/bash
# round 6, visit B: matcher GPU tests, per-call latencies, call timelines
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06_${1:-b}; rm -rf $O; mkdir -p $O
T0=$(date +%s); lap() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
timeout 900 python3 -m pytest tests/test_gpu_matcher.py tests/test_gpu_geometry.py tests/test_gpu_adapter_vs_reference.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
lap suite
timeout 300 python tools/latency_calls.py 100 $O/latency_calls.json > $O/latency_calls.txt 2>&1; echo "latency rc=$?"; cat $O/latency_calls.txt | tail -16
lap latency
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python tools/latency_calls.py 30 > $O/latency_under_trace.txt 2>&1
DB=$(find $O/kt -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
python tools/call_timeline.py $DB > $O/call_timelines.txt 2>&1
python tools/rocprof_summary.py $DB $O/kernel_stats.csv > /dev/null
cat $O/call_timelines.txt | grep -v "^      median" | head -120
rm -rf $O/kt
lap timelines
