cd /root/repo
export TMPDIR=/tmp
for p in 0 1; do
ORBX_STREAM_PRIO=$p timeout 300 python bench.py --workload kitti --steps 10 --warmup 2 --cpu-frames 0 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('prio=$p kitti', j['value'], j['ms_per_step'])"
done
ORBX_STREAM_PRIO=0 timeout 300 python bench.py --steps 20 --warmup 3 --cpu-frames 0 --no-profile 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('prio=0 euroc', j['value'], j['ms_per_step'])"
