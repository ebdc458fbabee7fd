#!/bin/bash
# experiment: one GPU, the 256-frame step as N host threads x (256 / N) frames, each thread its own extractor and streams (bench.py --threads --share-gpus shape)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/${1:-sub}; mkdir -p $O
for cfg in "1 256" "2 128" "4 64" "1 128"; do set -- $cfg
  timeout 200 python3 bench.py --threads --gpus $1 --batch $2 --steps 40 --warmup 5 --verify 0 2>/dev/null | tail -1 | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('threads $1 x $2 frames |', d['value'], 'kfeatures/s | ms per thread-step', d['ms_per_step'])"
done 2>&1 | tee $O/subbatch.log
