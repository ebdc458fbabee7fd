#!/bin/bash
# kernel timelines of the host-pointer matcher calls (rocprofv3 --kernel-trace over tools/latency_calls.py)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06_calls_${1:-a}; rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python tools/latency_calls.py 30 > $O/latency_under_trace.txt 2>&1
DB=$(find $O/kt -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2)
python tools/call_timeline.py $DB > $O/call_timelines.txt 2>&1
python tools/rocprof_summary.py $DB $O/kernel_stats.csv > /dev/null
cat $O/call_timelines.txt | head -150
rm -rf $O/kt
