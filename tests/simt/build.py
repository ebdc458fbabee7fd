#!/usr/bin/env python3
"""Builds tests/simt/build/liborbx_emul.so: liborbx's host and device sources (orb_slam3_amd/csrc, copied with a few textual
rewrites of what a CPU compiler cannot take: inline gfx950 assembly, `extern __shared__`, one hand-made LDS address) compiled by
the host clang++ against the SIMT emulator and the stand-in HIP runtime of this directory.  Test infrastructure only."""
import re
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE.parent.parent / "orb_slam3_amd" / "csrc"
BUILD = HERE / "build"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

REWRITES = [
    # dynamic LDS: every kernel's `extern __shared__ ... name[]` becomes a pointer to the emulator's LDS buffer
    (re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?(\w+)\s+(\w+)\[\];"), r"\1 *\2 = reinterpret_cast<\1 *>(simt::dyn_lds());"),
    # inline gfx950 assembly
    (re.compile(r'asm\("v_min3_i32 %0, %1, %2, %3"\s*:\s*"=v"\((\w+)\)\s*:\s*"v"\((\w+)\),\s*"v"\((\w+)\),\s*"v"\((\w+)\)\);'), r"\1 = std::min(\2, std::min(\3, \4));"),
    (re.compile(r'asm\("v_max3_i32 %0, %1, %2, %3"\s*:\s*"=v"\((\w+)\)\s*:\s*"v"\((\w+)\),\s*"v"\((\w+)\),\s*"v"\((\w+)\)\);'), r"\1 = std::max(\2, std::max(\3, \4));"),
    (re.compile(r'asm\("v_pk_min_u16 %0, %1, %2"\s*:\s*"=v"\((\w+)\)\s*:\s*"v"\((\w+)\),\s*"v"\((\w+)\)\);'), r"\1 = simt_pk_min_u16(\2, \3);"),
    (re.compile(r'asm\("v_pk_minimum3_f16 %0, %1, %2, %3"\s*:\s*"=v"\((\w+)\)\s*:\s*"v"\((\w+)\),\s*"v"\((\w+)\),\s*"v"\((\w+)\)\);'), r"\1 = simt_pk_min3_u16(\2, \3, \4);"),
    (re.compile(r'asm\("v_pk_maximum3_f16 %0, %1, %2, %3"\s*:\s*"=v"\((\w+)\)\s*:\s*"v"\((\w+)\),\s*"v"\((\w+)\),\s*"v"\((\w+)\)\);'), r"\1 = simt_pk_max3_u16(\2, \3, \4);"),
    (re.compile(r'asm volatile\(""\s*::[^;]*\);'), ";"),
    # k_blur_stream: hand-issued loads (SGPR base + 32-bit lane offset) and hand-counted waits
    (re.compile(r'asm volatile\("global_load_dword %0, %1, %2"\s*:\s*"=v"\(([\w.]+)\)\s*:\s*"v"\((\w+)\),\s*"s"\((\w+)\)\s*:\s*"memory"\);'),
     r"\1 = *reinterpret_cast<const uint32_t *>(\3 + \2);"),
    (re.compile(r'asm volatile\("s_waitcnt vmcnt\(\d+\)"\s*:[^;]*\);'), ";"),
    # k_describe stores row 31 of the orientation patch into what becomes row 0 of the BRIEF patch and relies on the wave's LDS
    # instructions executing in program order ACROSS lanes (lock step); the emulator's lanes are not in lock step: order the two phases
    (re.compile(r"(\n\s*)if \(c < 10\) \{(\s*)uint8_t \*d = Bp \+"), r"\1__builtin_amdgcn_wave_barrier();\1if (c < 10) {\2uint8_t *d = Bp +"),
    # k_octree_par1's chunked scatter: all lanes read the running offset hist[slot], then all lanes bump it (lock step); see simt::defer_add
    (re.compile(r"atomicAdd\(&hist\[slot\], 1u\);(\s*// LDS/memory operations of a wave are performed in order)"), r"simt::defer_add(&hist[slot], 1u);\1"),
    # k_describe's hand-made 32-bit LDS address
    (re.compile(r"\*reinterpret_cast<const __attribute__\(\(address_space\(3\)\)\) uint8_t \*>\((\w+)\)"), r"*simt::lds_ptr(\1)"),
]


def main():
    BUILD.mkdir(exist_ok=True)
    n_rew = 0
    for src in list(CSRC.glob("*.h")) + list(CSRC.glob("*.hip")) + list(CSRC.glob("*.inc")):
        text = src.read_text()
        for pat, rep in REWRITES:
            text, n = pat.subn(rep, text)
            n_rew += n
        text = text.replace('"../../include/orbx.h"', f'"{HERE.parent.parent / "include" / "orbx.h"}"')
        dst = BUILD / (src.name.replace(".hip", ".cc") if src.suffix == ".hip" else src.name)
        dst.write_text(text)
    if "asm(" in "".join((BUILD / f).read_text() for f in ("extractor_kernels.hip.h", "matcher_kernels.hip.h")):
        print("warning: inline assembly left in the copies", file=sys.stderr)
    asan = "--asan" in sys.argv[1:]   # AddressSanitizer build: device-side out-of-bounds accesses to "device" (heap) buffers are reported
    units = [a for a in sys.argv[1:] if not a.startswith("--")] or ["orbx_extractor.cc", "orbx_matcher.cc"]
    # UBSan build: conversions of out-of-range floats to integers (x86 and gfx950 give DIFFERENT results for those), shifts by >= the
    # width, signed overflow, out-of-bounds indices of fixed-size arrays; unaligned accesses are intended (the kernels rely on them)
    ubsan = "--ubsan" in sys.argv[1:]
    out = BUILD / ("liborbx_emul_asan.so" if asan else "liborbx_emul_ubsan.so" if ubsan else "liborbx_emul.so")
    cmd = [CLANG, "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-fno-strict-aliasing", "-Wno-unused-value",
           "-Wno-ignored-attributes", "-Wno-unknown-attributes", f"-I{HERE}", f"-I{BUILD}", "-o", str(out),
           str(HERE / "launch.cc")] + [str(BUILD / u) for u in units] + ["-ldl"]
    if asan:
        cmd[1:1] = ["-fsanitize=address", "-fno-omit-frame-pointer", "-shared-libasan"]
    if ubsan:
        rt = subprocess.run([CLANG, "-print-file-name=libclang_rt.ubsan_standalone-x86_64.so"], capture_output=True, text=True).stdout.strip()
        cmd[1:1] = ["-fsanitize=float-cast-overflow,shift,signed-integer-overflow,bounds,integer-divide-by-zero,float-divide-by-zero",
                    "-fno-omit-frame-pointer", "-shared-libsan", f"-Wl,-rpath,{Path(rt).parent}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    sys.stderr.write(r.stderr[-6000:])
    print("rewrites", n_rew, "rc", r.returncode)
    if r.returncode == 0 and not asan and not ubsan:   # the quad-tree kernels alone (octree_emul.cc): 0.1 s per level, for sweeps
        r = subprocess.run([CLANG, "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-value", "-Wno-ignored-attributes",
                            f"-I{HERE}", f"-I{BUILD}", "-o", str(BUILD / "liboctree_emul.so"), str(HERE / "octree_emul.cc"), str(HERE / "launch.cc")],
                           capture_output=True, text=True)
        sys.stderr.write(r.stderr[-3000:])
    return r.returncode


if __name__ == "__main__":
    sys.exit(main())
