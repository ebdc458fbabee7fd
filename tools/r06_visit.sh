#!/bin/bash
# round 6, visit A: GPU suite, per-call latencies with both transports (k_xfer launches vs the DMA engine), then the driver's bench line
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06_${1:-a}; rm -rf $O; mkdir -p $O
T0=$(date +%s); lap() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
lap suite
timeout 300 python tools/latency_calls.py 100 $O/latency_calls_kernel_xfer.json > $O/latency_calls_kernel_xfer.txt 2>&1; echo "latency (k_xfer) rc=$?"; cat $O/latency_calls_kernel_xfer.txt | tail -16
lap latency-kxfer
ORBX_MATCHER_DMA=1 timeout 300 python tools/latency_calls.py 100 $O/latency_calls_dma.json > $O/latency_calls_dma.txt 2>&1; echo "latency (DMA) rc=$?"; cat $O/latency_calls_dma.txt | tail -16
lap latency-dma
( time timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err | tail -1 > $O/bench_euroc.json ) 2>&1 | grep real
cut -c1-400 $O/bench_euroc.json; echo; tail -5 $O/bench.err
lap bench
python3 -c "import cv2; print('cv2', cv2.__version__)" 2>/dev/null && python3 tools/gen_ocv_golden.py && echo "OpenCV goldens written" || echo "cv2: not importable on this box"
