#!/usr/bin/env python3
"""bench.py's latency.calls_per_entry_point block on its own: every host-pointer matcher entry point, one call at a time, beside the CPU oracle's
time for the identical call (parity asserted).  ORBX_MATCHER_DMA=1 selects round 5's transport (DMA engine) for an A/B.
usage: python tools/latency_calls.py [calls per entry point] [out.json]"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import orb_slam3_amd as osa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rows = bench.latency_calls(osa, 0, n)
for e in rows:
    print(f"{e['median_ms'] * 1e3:8.1f} us  p90 {e['p90_ms'] * 1e3:8.1f}  min {e['min_ms'] * 1e3:8.1f} | cpu {e['cpu_oracle_median_ms'] * 1e3:8.1f} us | x{e['speedup_vs_cpu_oracle']:5.2f} | "
          f"{'ok ' if e['parity_checked'] else 'BAD'} | {e['call'][:110]}")
if len(sys.argv) > 2:
    Path(sys.argv[2]).write_text(json.dumps(rows, indent=1))
sys.exit(0 if all(e["parity_checked"] for e in rows) else 1)
