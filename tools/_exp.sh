cd /root/repo
export TMPDIR=/tmp
for R in 4 1 2; do
cp gpurun_liborbx_R$R.so orb_slam3_amd/liborbx.so
echo "R=$R"
timeout 300 python bench.py --steps 20 --warmup 3 --cpu-frames 0 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['kernels']['k_pyr_resize'])"
done
timeout 900 python -m pytest tests/test_gpu_extractor.py -x -q 2>&1 | tail -3
