cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_extractor.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 3 --cpu-frames 0 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], {k:v['avg_ms'] for k,v in j['kernels'].items()})"
