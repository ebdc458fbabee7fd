#!/bin/bash
# First GPU visit of round 3 (prepared at the end of round 2, when the GPU budget was spent):
#  1. the two discriminating reproductions of the open 1007-vs-1008 keypoint difference + the stage-wise image sweep (xfail in round 2; green with the fix at the end of round 2, ordinary tests now),
#     with full output: which of them fails, and at which stage, says whether it is data or ordering
#     + tools/open_item_probe.sh: failure rate of the short-batch loop with side streams off / serialized kernels / blocking launches
#  2. the opt-in pipeline test of every round-2 switch (child processes)
#  3. A/B of the kernels written under the emulator only: k_pyr_chain<1|2>, k_grid_build2, k_window_best2<DPP>, k_describe2, k_octree_par_t
# usage: gpurun --timeout 600 -- 'bash tools/r03_first.sh'
mkdir -p gpurun_out/r03a
export TMPDIR=/tmp
timeout 200 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_extractor.py -m gpu -q -rxX \
    -k "open_small_canvas or open_random_image_sweep or open_first_host_batch" > gpurun_out/r03a/open.log 2>&1
echo "open items rc=$?"; tail -40 gpurun_out/r03a/open.log
# the same sequence with the clear of DevBuf::ensure left unsynchronised (the code before the fix): expected to FAIL (xfail) if the
# asynchronous hipMemset is the cause
ORBX_ENSURE_NOSYNC=1 timeout 120 python3 -m pytest tests/test_gpu_pipeline.py -m gpu -q -rxX -k "open_first_host_batch" > gpurun_out/r03a/open_nosync.log 2>&1
echo "unsynchronised clear rc=$?"; tail -15 gpurun_out/r03a/open_nosync.log
ORBX_PROBE_SETTINGS="ORBX_NONE=1|ORBX_ENSURE_NOSYNC=1" bash tools/open_item_probe.sh 5 2>&1 | tee gpurun_out/r03a/probe.log
ORBX_TEST_SWITCHES=1 timeout 240 python3 -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k alternative_switches > gpurun_out/r03a/switches.log 2>&1
echo "switches rc=$?"; tail -12 gpurun_out/r03a/switches.log
bash tools/ab.sh "ORBX_NONE=1" "ORBX_PYR_CHAIN=1" "ORBX_PYR_CHAIN=2" "ORBX_GRID_BUILD=2" "ORBX_WINDOW_DPP=1" "ORBX_DESCRIBE=2" "ORBX_OCTREE_KEYS=2048" "ORBX_PYR_CHAIN=1 ORBX_GRID_BUILD=2 ORBX_WINDOW_DPP=1" 2>&1 | tee gpurun_out/r03a/ab.log
ORBX_PYR_CHAIN=1 ORBX_GRID_BUILD=2 ORBX_WINDOW_DPP=1 bash tools/quick_prof.sh > /dev/null 2>&1; head -14 gpurun_out/qp/stats.csv | tee gpurun_out/r03a/prof_chain.log
