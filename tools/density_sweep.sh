#!/bin/bash
# Candidate-density sweep: the EuRoC step on scenes between the quad scene (1 x) and the texture scene (13 x the FAST candidates):
# bench.py --scene blend:<a>, whole-step parity on, per-kernel profile on; one JSON with ms_per_step, candidates per frame, stage stats and kernel times per density.
# usage: bash tools/density_sweep.sh <tag>   -> gpurun_out/density_<tag>/density_sweep.json (+ the bench lines)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-a}
O=gpurun_out/density_$TAG; rm -rf $O; mkdir -p $O
for sc in quads blend:0.1 blend:0.2 blend:0.35 blend:0.5 blend:0.75 texture; do
  f=$O/bench_$(echo $sc | tr ':.' '__').json
  timeout 400 python3 bench.py --scene $sc --cpu-frames 0 --no-pmc --no-other-workloads --latency 0 --repeat 2 --verify 8 2> $O/err.txt | tail -1 > $f
  python3 - "$sc" "$f" <<'PY'
import json, sys
sc, f = sys.argv[1:3]
try:
    d = json.load(open(f))
except Exception as e:
    print(sc, "| FAILED", e); sys.exit(0)
st = d["stage_stats_last_step"]
k = {n: round(v["avg_ms"] * 1e3, 1) for n, v in (d.get("kernels") or {}).items()}
print(f"{sc:12s} | ms/step {d['ms_per_step']:6.3f} | Mfeat/s {d['value'] / 1e3:7.1f} | parity {d.get('parity_checked')} | cand/frame {st.get('fast_candidates', 0) // 256} | {st} | us {k}")
PY
done 2>&1 | tee $O/density_sweep.txt
python3 - $O <<'PY'
import json, sys, glob, os
O = sys.argv[1]
rows = []
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception:
        continue
    rows.append({"scene": os.path.basename(f)[6:-5], "ms_per_step": d["ms_per_step"], "value_kfeatures_s": d["value"], "parity_checked": d.get("parity_checked"),
                 "stage_stats_last_step": d["stage_stats_last_step"], "fast_queues": d["fast_queues"]["in_force"],
                 "kernel_us": {n: round(v["avg_ms"] * 1e3, 1) for n, v in (d.get("kernels") or {}).items()}})
json.dump(rows, open(O + "/density_sweep.json", "w"), indent=1)
PY
