#!/bin/bash
# SQ instruction / cycle counters per kernel for the current build (one rocprofv3 --pmc pass)
export TMPDIR=/tmp
O=gpurun_out/sq; rm -rf $O; mkdir -p $O
env "$@" timeout 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O -o sq -- python bench.py --pmc-child --workload ${WL:-euroc} --steps 3 --warmup 1 > /dev/null 2>&1
python tools/pmc_sq.py sqq $(find $O -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2) > /dev/null; mv profiles/sqq_pmc_sq_counters.csv $O/; cut -d, -f1,4,5,6,8,9,10,11 $O/sqq_pmc_sq_counters.csv
