#!/bin/bash
# build liborbx.so and the emulator library from anywhere (absolute paths), then run the given command from the repo root
set -e
R=/root/repo
[ -d "$R" ] || R=$(cd "$(dirname "$0")/.." && pwd)
make -C $R/orb_slam3_amd/csrc 2>&1 | grep -i "error" -A6 || true
python3 $R/tests/simt/build.py 2>&1 | tail -1
cd $R
if [ $# -gt 0 ]; then "$@"; fi
