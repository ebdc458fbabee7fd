cd /root/repo
export TMPDIR=/tmp
rm -rf gpurun_out/prof_d; mkdir -p gpurun_out/prof_d
ORBX_SIDE_STREAMS=0 timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_d -o disp -- python bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-profile > /dev/null 2>&1
python tools/rocprof_dispatches.py $(find gpurun_out/prof_d -name "*.db" | head -1) | grep -v copyBuffer
