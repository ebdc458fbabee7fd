cd /root/repo
export TMPDIR=/tmp
for b in 128 256 512; do
timeout 600 python bench.py --steps 16 --warmup 3 --cpu-frames 0 --no-profile --batch $b 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('B=$b', j['value'], j['ms_per_step'])"
done
