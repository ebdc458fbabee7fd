run() { (cd $1 && env $3 python3 bench.py --workload $4 --steps 30 --warmup 5 --cpu-frames 0 --no-profile --verify 0 $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$4', '$3', d['value'], d['ms_per_step'])"); }
for v in "ORBX_DUMMY_STREAMS=0" "ORBX_DUMMY_STREAMS=1" "ORBX_DUMMY_STREAMS=2" "ORBX_DUMMY_STREAMS=3"; do run . "--repeat 1 --latency 0" "$v" kitti; done
for v in "ORBX_DUMMY_STREAMS=1" "ORBX_DUMMY_STREAMS=2" "ORBX_DUMMY_STREAMS=3"; do run . "--repeat 1 --latency 0" "$v" euroc; done
