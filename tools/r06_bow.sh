#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for d in 0 1 2 3; do echo "ORBX_BOW_DEBUG=$d"; ORBX_BOW_DEBUG=$d python tools/bow_probe.py 2>&1 | tail -1; done
