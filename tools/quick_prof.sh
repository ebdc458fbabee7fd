#!/bin/bash
# serialized per-kernel durations (rocprofv3 --kernel-trace --stats) of a few extract+match steps; optional SQ counters
export TMPDIR=/tmp
O=gpurun_out/qp; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/se -o se -- python bench.py --pmc-child --workload ${1:-euroc} --steps 8 --warmup 2 > /dev/null 2>&1
python tools/rocprof_summary.py $(find $O/se -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2) $O/stats.csv
if [ -n "${2:-}" ]; then
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/sq -o sq -- python bench.py --pmc-child --workload ${1:-euroc} --steps 3 --warmup 1 > /dev/null 2>&1
python tools/pmc_sq.py qp $(find $O/sq -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2); mv profiles/qp_pmc_sq_counters.csv $O/
fi
