#!/bin/bash
# end-of-round evidence: GPU test-suite, the driver's bench line (with the KITTI / TUM-VI child runs embedded), serialized kernel stats, overlapped kernel
# stats, the pipelined timeline, single-frame latency, SQ counters, PMC traffic, the two other workloads standalone, the bench parity with the blur pass
# forced -- in THIS order: most important first, every step under its own timeout, so that a visit cut short keeps what it has.
# usage: bash tools/final_round.sh <tag>   -> gpurun_out/measure_<tag>/profiles_copy/
set -u
TAG=${1:-r06_a}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/measure_$TAG
rm -rf $O; mkdir -p $O/profiles_copy profiles
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $1"; }
biggest_db() { find "$1" -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2; }
CHILD="python bench.py --pmc-child --workload euroc"
timeout 600 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log; tail -3 $O/pytest.log > $O/profiles_copy/${TAG}_gpu_suite.log
lap suite
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err | tail -1 > $O/profiles_copy/${TAG}_bench_euroc.json ) 2>&1 | grep real
cut -c1-300 $O/profiles_copy/${TAG}_bench_euroc.json; echo
lap bench
timeout 60 rocprofv3 --kernel-trace --stats -d $O/se -o se -- $CHILD --steps 12 --warmup 3 > /dev/null 2>&1
python tools/rocprof_summary.py $(biggest_db $O/se) $O/profiles_copy/${TAG}_serialized_kernel_stats.csv
python3 tools/rocprof_dispatches.py $(biggest_db $O/se) | grep -i "pyr\|strip\|describe" > $O/profiles_copy/${TAG}_dispatches.txt
lap serialized
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 timeout 60 rocprofv3 --kernel-trace --stats -d $O/ov -o ov -- python bench.py --steps 12 --warmup 3 --cpu-frames 0 --no-profile --verify 0 --no-pmc --repeat 1 --latency 0 > /dev/null 2>&1
python tools/rocprof_summary.py $(biggest_db $O/ov) $O/profiles_copy/${TAG}_overlapped_kernel_stats.csv > /dev/null
bash tools/timeline.sh > $O/profiles_copy/${TAG}_timeline.txt 2>&1
lap overlapped+timeline
timeout 120 python tools/latency.py > $O/profiles_copy/${TAG}_latency.txt 2>&1
timeout 60 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/sq -o sq -- $CHILD --steps 3 --warmup 1 > /dev/null 2>&1
python tools/pmc_sq.py $TAG $(biggest_db $O/sq) > /dev/null && cp profiles/${TAG}_pmc_sq_counters.csv $O/profiles_copy/
timeout 60 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/pf -o pf -- $CHILD --steps 3 --warmup 1 > /dev/null 2>&1
timeout 60 rocprofv3 --pmc WRITE_SIZE -d $O/pw -o pw -- $CHILD --steps 3 --warmup 1 > /dev/null 2>&1
python tools/pmc_summary.py $(biggest_db $O/pf) $(biggest_db $O/pw) $TAG && cp profiles/${TAG}_pmc_traffic.csv $O/profiles_copy/
lap counters
for wl in tumvi kitti; do
  timeout 200 python bench.py --workload $wl --cpu-frames 48 2> $O/bench_$wl.err | tail -1 > $O/profiles_copy/${TAG}_bench_$wl.json
  cut -c1-200 $O/profiles_copy/${TAG}_bench_$wl.json; echo
done
lap workloads
# the blur pass (k_blur_stream + k_describe) is off the default path of all three geometries -- one whole-step parity check with it forced
ORBX_FUSED_BLUR=0 timeout 120 python bench.py --workload euroc --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 --no-profile --verify 8 --repeat 1 2> /dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ORBX_FUSED_BLUR=0 euroc: ms_per_step', d['ms_per_step'], 'parity_checked', d['parity_checked'])" > $O/profiles_copy/${TAG}_blur_pass_forced_parity.txt 2>&1
cat $O/profiles_copy/${TAG}_blur_pass_forced_parity.txt
# round 6: every host-pointer matcher entry point one call at a time (both transports), the kernel timeline of one call of each, the candidate-density sweep,
# and the N-rank path with two ranks on the one GPU (both clocks, parity on both ranks: VERDICT r5 item 9)
timeout 300 python tools/latency_calls.py 100 > $O/profiles_copy/${TAG}_latency_calls_kernel_xfer.txt 2>&1
ORBX_MATCHER_DMA=1 timeout 300 python tools/latency_calls.py 100 > $O/profiles_copy/${TAG}_latency_calls_dma_engine.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python tools/latency_calls.py 30 > /dev/null 2>&1
python tools/call_timeline.py $(biggest_db $O/kt) > $O/profiles_copy/${TAG}_call_timelines.txt 2>&1
lap calls
bash tools/density_sweep.sh $TAG > /dev/null 2>&1; cp gpurun_out/density_$TAG/density_sweep.txt $O/profiles_copy/${TAG}_density_sweep.txt; cp gpurun_out/density_$TAG/density_sweep.json $O/profiles_copy/${TAG}_density_sweep.json
tail -7 $O/profiles_copy/${TAG}_density_sweep.txt | cut -c1-120
lap density
timeout 300 python bench.py --gpus 2 --share-gpus --steps 20 --warmup 5 --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 --no-profile 2> $O/two_ranks.err | tail -1 > $O/profiles_copy/${TAG}_two_ranks_one_gpu.json
python3 -c "
import json
d = json.load(open('$O/profiles_copy/${TAG}_two_ranks_one_gpu.json')); print('2 ranks on one GPU: n_gpus', d['n_gpus'], 'value', d['value'], 'per_rank', [(r.get('rank'), r.get('parity_checked') is not None) for r in d.get('per_rank', [])], 'pcie', d['pcie_inclusive']['value'])"
lap two-ranks
# steady state: one region of K = 4000 pipelined steps (the contract's K = 20 regions pay the pipeline's fill and drain); SOAK=0 skips it
if [ "${SOAK:-1}" != "0" ]; then
  timeout 150 python bench.py --gpus 1 --steps 4000 --warmup 5 --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 --no-profile --repeat 2 2> /dev/null | tail -1 > $O/profiles_copy/${TAG}_soak_4000_step_regions.json
  python3 -c "
import sys, json
d = json.load(open('$O/profiles_copy/${TAG}_soak_4000_step_regions.json')); print('K = 4000: ms_per_step', d['ms_per_step'], d['repeats']['ms_per_step_in_order'], 'value', d['value'], 'parity', d['parity_checked'], 'pcie-inclusive', d['pcie_inclusive']['ms_per_step'])"
  lap soak
fi
rm -rf $O/se $O/pf $O/pw $O/sq $O/ov $O/kt
ls $O/profiles_copy
