#!/usr/bin/env python3
"""Which kernels' code objects changed between a commit and the working tree?  Compiles both versions of orb_slam3_amd/csrc to gfx950 ISA
(hipcc --cuda-device-only -S, no GPU needed) and compares the opcode histogram of every kernel.  Used at the end of round 2, when
changes could no longer be validated on hardware, to show that the default kernels are the ones the last GPU run validated.
usage: python tools/isa_diff.py <commit>"""
import collections
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", "--cuda-device-only", "-S"]


def hist(asm: Path):
    out, cur = {}, None
    for ln in asm.read_text().split("\n"):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            out[cur] = collections.Counter()
            continue
        m = re.match(r"^\t([a-z_0-9]+)", ln)
        if cur and m and not m.group(1).startswith("."):
            out[cur][m.group(1)] += 1
    return {k: v for k, v in out.items() if v}


def compile_tree(csrc: Path, td: Path, tag: str):
    res = {}
    for unit in ("orbx_extractor", "orbx_matcher"):
        out = td / f"{tag}_{unit}.s"
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-o", str(out), f"{unit}.hip"], check=True, capture_output=True, cwd=csrc)
        res.update(hist(out))
    return res


def main():
    commit = sys.argv[1]
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        subprocess.run(f"git archive {commit} orb_slam3_amd/csrc include | tar -x -C {td}", shell=True, check=True, cwd=ROOT)
        old = compile_tree(td / "orb_slam3_amd" / "csrc", td, "old")
        new = compile_tree(ROOT / "orb_slam3_amd" / "csrc", td, "new")
    same = [k for k in old if k in new and old[k] == new[k]]
    print(f"{len(same)} kernels identical")
    for k in old:
        if k in new and old[k] != new[k]:
            d = {op: (old[k][op], new[k][op]) for op in set(old[k]) | set(new[k]) if old[k][op] != new[k][op]}
            print("changed:", k[:100], d)
        elif k not in new:
            print("removed / renamed:", k[:100])
    for k in new:
        if k not in old:
            print("added:", k[:100])


if __name__ == "__main__":
    main()
