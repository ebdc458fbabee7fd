#!/bin/bash
# GPU visit: the whole GPU test-suite, then the full default bench line (all new blocks: repeats, full-batch parity, latency, other workloads, PMC
# traffic) with k_fast_strip and, for the decision between the two FAST forms, with the per-cell k_fast_ini
mkdir -p gpurun_out/r03d
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03d/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r03d/pytest.log
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03d/bench_strip.json 2> gpurun_out/r03d/bench_strip.err ) 2>&1 | grep real
echo "bench strip rc=$?"; tail -c 600 gpurun_out/r03d/bench_strip.err; cut -c1-1500 gpurun_out/r03d/bench_strip.json
( time ORBX_FAST_STRIP=0 timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 48 > gpurun_out/r03d/bench_ini.json 2> gpurun_out/r03d/bench_ini.err ) 2>&1 | grep real
echo "bench ini rc=$?"; tail -c 300 gpurun_out/r03d/bench_ini.err
python3 - <<'PY'
import json
for n in ("strip","ini"):
    try:
        d=json.loads(open(f"gpurun_out/r03d/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["repeats"]["ms_per_step"], "pcie", d["value_pcie_inclusive"], "roofline", {k:d["roofline"].get(k) for k in ("kernel","achieved","frac","traffic","avg_launch_ms","extract_all_kernels_frac")}, d["roofline"].get("traffic_detail",{}).get("traffic_over_algorithmic"))
        print("  latency", d["latency"]); print("  parity", d["parity_checked"])
        for k,v in (d["other_workloads"] or {}).items(): print("  ", k, {a:v.get(a) for a in ("value","ms_per_step","parity_checked","error")})
        print("  kernels", {k:v["avg_ms"] for k,v in d["kernels"].items()})
    except Exception as e: print(n, "ERR", e)
PY
