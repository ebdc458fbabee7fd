cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_extractor.py -x -q 2>&1 | tail -15
for m in par; do
ORBX_OCTREE=$m timeout 300 python bench.py --steps 20 --warmup 3 --cpu-frames 0 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['kernels']['k_octree'])"
done
mkdir -p gpurun_out/prof_d
ORBX_SIDE_STREAMS=0 timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_d -o disp -- python bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-profile > /dev/null 2>&1
python tools/rocprof_dispatches.py $(find gpurun_out/prof_d -name "*.db" | head -1) | grep -v copyBuffer
