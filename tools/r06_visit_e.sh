#!/bin/bash
# round 6, visit e: matcher / geometry / adapter GPU tests with the fisheye triangulation kernel, per-call latencies, call timelines
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06_${1:-i}; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_matcher.py tests/test_gpu_geometry.py tests/test_gpu_adapter_vs_reference.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 300 python tools/latency_calls.py 100 > $O/latency_calls.txt 2>&1; cut -c1-200 $O/latency_calls.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python tools/latency_calls.py 30 > /dev/null 2>&1
python tools/call_timeline.py $(find $O/kt -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2) > $O/call_timelines.txt 2>&1
rm -rf $O/kt
grep -B1 -A12 "k_tri_kb8" $O/call_timelines.txt | head -40
