#!/bin/bash
# the texture stress scene with the FAST queues tuned by orbx_tune_fast_queues (bench.py does it after the settle steps), the GPU test of the tuning, the default line beside it
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/${1:-tex}; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_extractor.py -m gpu -x -q -k "tuning or texture" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for sc in texture quads; do
  timeout 300 python3 bench.py --scene $sc --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 --repeat 3 2>$O/err_$sc.txt | tail -1 > $O/bench_$sc.json
  python3 -c "import json; d=json.load(open('$O/bench_$sc.json')); print('$sc |', d['value'], '| ms', d['ms_per_step'], '| parity', d.get('parity_checked'), '|', d['fast_queues']['in_force'], '|', d['stage_stats_last_step'], '|', (d.get('profile') or {}))"
done 2>&1 | tee $O/texture.log
