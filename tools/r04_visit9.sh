#!/bin/bash
# round 4, GPU visit 9: the host-input clock with the upload split over four streams; pipeline tests at the bench's shape
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do python3 bench.py --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 --no-profile 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['pcie_inclusive']; print('resident', d['value'], d['ms_per_step'], '| host-input', p['value'], p['ms_per_step'], p['h2d_GBs'], bool(p['parity_checked']))"; done
python3 tools/microbench/h2d_streams.py
