cd /root/repo
export TMPDIR=/tmp
for v in "" "ORBX_BENCH_SKIP_MATCH=1" "ORBX_SIDE_STREAMS=0" "ORBX_BLUR_SIDE=0 ORBX_BENCH_SKIP_MATCH=1"; do
env $v timeout 300 python bench.py --steps 30 --warmup 3 --cpu-frames 0 --no-profile 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$v', j['value'], j['ms_per_step'])"
done
