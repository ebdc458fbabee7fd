#!/bin/bash
# The emulator battery of round 2, in one go (CPU only, about 45 minutes on 6 cores).  Every line runs the C-ABI GPU tests (or the
# pipelined loop / the switch matrix / bench.py) against liborbx's own sources compiled for the SIMT emulator, under one family of
# adversarial settings (tests/simt/README.md).  Expected: every run ends with the same three failures and nothing else --
#   test_cpp_adapter_end_to_end, test_alternative_kernel_paths   (compile against / spawn children with the real liborbx.so)
#   test_async_entry_points_refuse_pageable_host_memory          (every pointer counts as pinned under the emulator)
# usage: bash tests/simt/battery.sh [workers, default 6]      logs: /tmp/simt_battery/*.log
cd "$(dirname "$0")/../.." || exit 1
N=${1:-6}
OUT=/tmp/simt_battery; mkdir -p $OUT
python tests/simt/build.py && python tests/simt/build.py --ubsan && python tests/simt/build.py --asan || exit 1
ASAN=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
T="tests/test_gpu_matcher.py tests/test_gpu_geometry.py tests/test_gpu_extractor.py tests/test_gpu_pipeline.py"
SKIP="--deselect tests/test_gpu_pipeline.py::test_bench_shape_pipeline_alternating_inputs --deselect tests/test_gpu_pipeline.py::test_open_first_host_batch_into_fresh_slab"
run() {   # name, environment assignments...
  local name=$1; shift
  env "$@" timeout 3400 python -m pytest $T -q -m gpu -n $N -p no:cacheprovider $SKIP > $OUT/$name.log 2>&1
  echo "$name: $(tail -1 $OUT/$name.log)"
}
run order_garbage   ORBX_TEST_EMULATOR=1 SIMT_STRICT_LANES=1 SIMT_MALLOC_FILL=r2 SIMT_BLOCK_ORDER=5 SIMT_LANE_ORDER=3
run lds_shuffle     ORBX_TEST_EMULATOR=1 SIMT_LDS_RANDOM=11 SIMT_SHUFFLE=4
run ubsan           ORBX_TEST_EMULATOR=ubsan UBSAN_OPTIONS=log_path=$OUT/ubsan_report
run streams         ORBX_TEST_EMULATOR=1 SIMT_STREAM_FUZZ=41 SIMT_KERNEL_SPLIT=8 SIMT_MEMSET_ASYNC=1
run streams_first   ORBX_TEST_EMULATOR=1 SIMT_STREAM_FUZZ=first SIMT_MEMSET_ASYNC=1
run asan_streams    ORBX_TEST_EMULATOR=asan SIMT_STREAM_FUZZ=last SIMT_MEMSET_ASYNC=1 LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:log_path=$OUT/asan_report
echo "sanitizer reports: $(cat $OUT/ubsan_report.* $OUT/asan_report.* 2>/dev/null | grep -c 'runtime error\|ERROR: AddressSanitizer')"
# the 21 switches (incl. the kernels prepared for round 3) and the three bench lines, all shuffles and garbage on
ORBX_TEST_EMULATOR_FULL=1 SIMT_BLOCK_ORDER=3 SIMT_LANE_ORDER=4 SIMT_MALLOC_FILL=r6 SIMT_LDS_RANDOM=2 SIMT_STRICT_LANES=1 SIMT_SHUFFLE=9 \
  timeout 3400 python -m pytest tests/test_simt_emulation.py -q -n $N -k "switch_matrix or bench_line" -p no:cacheprovider > $OUT/matrix.log 2>&1
echo "matrix: $(tail -1 $OUT/matrix.log)"
# the reproduction of the round-2 hardware failure: fails with the unsynchronised clear, passes with the fix
for v in "ORBX_ENSURE_NOSYNC=1" "ORBX_NONE=1"; do
  env $v SIMT_STREAM_FUZZ=5 SIMT_MEMSET_ASYNC=1 python - <<'PY' > $OUT/memset_$v.log 2>&1
import os, subprocess, sys
sys.path.insert(0, "tests")
import test_simt_emulation as t
r = subprocess.run([sys.executable, "-c", t.PRELUDE + t.PIPELINE], capture_output=True, text=True)
print("pipelined loop:", "ok" if r.returncode == 0 else "FAILED " + r.stderr.strip().splitlines()[-1])
PY
  echo "$v: $(tail -1 $OUT/memset_$v.log)"
done
