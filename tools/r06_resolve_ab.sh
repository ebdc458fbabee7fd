#!/bin/bash
# A/B of the replay kernel of SearchByProjection (ORBX_RESOLVE_WAVES = 1: one-wave k_greedy_resolve_t, 2 / 4 / 8: k_resolve_wide_t): matcher GPU tests,
# serialized kernel time inside the batched step, the bench's step time, single-call latencies
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/resolve_ab; rm -rf $O; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_matcher.py tests/test_gpu_pipeline.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for w in 1 2 4 8; do
  export ORBX_RESOLVE_WAVES=$w
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/se$w -o se -- python bench.py --pmc-child --workload euroc --steps 8 --warmup 2 > /dev/null 2>&1
  python tools/rocprof_summary.py $(find $O/se$w -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2) $O/stats_w$w.csv | grep -i "resolve\|window\|grid" | sed "s/^/W=$w serialized: /"
  rm -rf $O/se$w
  timeout 200 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 --no-profile --repeat 3 2> /dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('W=$w bench: ms_per_step', d['ms_per_step'], d.get('repeats', {}).get('ms_per_step_in_order'), 'parity', d['parity_checked'])"
  timeout 300 python tools/latency_calls.py 60 2>&1 | grep -i "SearchByProjection\|BAD" | cut -c1-150 | sed "s/^/W=$w call: /"
done 2>&1 | tee $O/ab.txt
