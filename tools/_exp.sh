cd /root/repo
export TMPDIR=/tmp
for l in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 4 --cpu-frames 0 --no-profile --lanes $l --batch 128 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('lanes=$l', j['value'], j['ms_per_step'])"
done
