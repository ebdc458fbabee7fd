cd /root/repo
export TMPDIR=/tmp
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 3 --cpu-frames 0 --no-profile 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
done
ORBX_BENCH_SKIP_MATCH=1 timeout 300 python bench.py --steps 20 --warmup 3 --cpu-frames 0 --no-profile 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('nomatch', j['value'], j['ms_per_step'])"
