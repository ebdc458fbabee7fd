#!/usr/bin/env python3
"""Static per-kernel report from the gfx950 ISA of the current sources: instruction counts by class, VGPRs, occupancy, scratch.
usage: python tools/isa_static.py [out.csv]      (hipcc --cuda-device-only -S on both translation units; no GPU needed)"""
import collections
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "orb_slam3_amd" / "csrc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", "--cuda-device-only", "-S"]


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        return r.stdout.split("\n") if r.returncode == 0 else names
    except FileNotFoundError:
        return names


def main():
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for unit in ("orbx_extractor.hip", "orbx_matcher.hip"):
            out = Path(td) / (unit + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-o", str(out), str(CSRC / unit)], check=True, capture_output=True, cwd=CSRC)
            cur, stats, meta = None, {}, {}
            for ln in out.read_text().split("\n"):
                m = re.match(r"^(_Z\w+):", ln)
                if m:
                    cur = m.group(1)
                    stats[cur] = collections.Counter()
                    continue
                if cur is None:
                    continue
                m = re.match(r"^; (NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)", ln)
                if m:
                    meta.setdefault(cur, {})[m.group(1)] = int(m.group(2))
                    continue
                m = re.match(r"^\t([a-z_0-9]+)", ln)
                if m and not m.group(1).startswith("."):
                    op = m.group(1)
                    cls = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
                           else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
                    stats[cur][cls] += 1
            names = [k for k in stats if k in meta and "Occupancy" in meta[k]]
            for k, d in zip(names, demangle(names)):
                short = re.sub(r"\(.*", "", d).replace("orbx::", "").replace("void ", "")
                c, mt = stats[k], meta[k]
                rows.append((unit, short, c["valu"], c["salu"], c["lds"], c["vmem"], mt.get("NumVgprs", 0), mt.get("Occupancy", 0),
                             mt.get("LDSByteSize", 0), mt.get("ScratchSize", 0)))
    rows.sort()
    lines = ["unit,kernel,valu_static,salu_static,lds_static,vmem_static,vgprs,occupancy_waves_per_simd,static_lds_bytes,scratch_bytes"]
    lines += [",".join(str(x) if not isinstance(x, str) else f'"{x}"' for x in r) for r in rows]
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
