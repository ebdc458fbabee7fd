#!/usr/bin/env python3
"""Static checks on the gfx950 ISA hipcc emits for liborbx's kernels (no GPU needed: hipcc --cuda-device-only -S).

1. In-flight VMEM destinations.  k_blur_stream issues its row loads as `asm volatile("global_load_dword ...")` and retires them
   with a hand-counted `s_waitcnt vmcnt(12)`; the compiler's own wait insertion cannot see those loads.  check_vmem() walks the
   kernel's control-flow graph (every path, loops to a fixed point) with the FIFO of outstanding loads as the state and reports every
   instruction that reads or writes a VGPR whose load has not been retired by a wait.  Retirement rule (gfx9: loads return in
   order among loads; stores share the counter and can only make a wait stricter): after `s_waitcnt vmcnt(k)` a load is complete
   when at least k loads were issued after it.  The same walk holds for compiler-scheduled loads, so it runs over every kernel.
2. Resource budgets the design relies on (VGPRs -> waves per SIMD, static LDS, scratch).

usage: python tools/isa_vmem_check.py            (prints a report; exit code 1 on a violation)
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "orb_slam3_amd" / "csrc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", "-w", "--cuda-device-only", "-S"]
UNITS = ("orbx_extractor.hip", "orbx_matcher.hip")

_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
_LOAD = re.compile(r"^(global_load|buffer_load|flat_load|scratch_load|global_atomic|buffer_atomic|flat_atomic)")
_STORE = re.compile(r"^(global_store|buffer_store|flat_store|scratch_store)")


def compile_unit(unit, extra=()):
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "u.s"
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-o", str(out), str(CSRC / unit)], check=True, capture_output=True, cwd=CSRC)
        return out.read_text()


def regs_of(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.add(m.group(1) + m.group(2))
        else:
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add(m.group(3) + str(i))
    return out


def split_kernels(asm):
    """{mangled name: (list of (op, operand text), {label: index}, meta dict)} for every kernel (.amdhsa_kernel) of the unit"""
    kernels, cur, ins, labels, meta = {}, None, [], {}, {}
    is_kernel = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm, re.M))
    for ln in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1) if m.group(1) in is_kernel else None
            ins, labels, meta = [], {}, {}
            if cur:
                kernels[cur] = (ins, labels, meta)
            continue
        if cur is None:
            continue
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        m = re.match(r"^; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)", ln)
        if m:
            meta[m.group(1)] = int(m.group(2))
            continue
        if re.match(r"^\s*\.(end_amdhsa_kernel|section|text)", ln) and meta.get("Occupancy") is not None:
            cur = None
            continue
        m = re.match(r"^\t([a-z_0-9]+)\s*([^;]*)", ln)
        if m and not m.group(1).startswith("."):
            ins.append((m.group(1), m.group(2).strip()))
    return kernels


def check_vmem(ins, labels, count_stores=True):
    """-> (violations [(index, op, operands, register)], loads, vmcnt waits).
    Forward dataflow to a fixed point over the kernel's control-flow graph.  State at an instruction = {register: n}: the register is the
    destination of a load that may still be outstanding on some path, and on every such path at least n loads were issued after it.
    Join = union of the registers with the smaller n (the conservative side: a wait vmcnt(k) retires a load only when n >= k).
    count_stores=True is the model of LLVM's own wait insertion on gfx9 (every VMEM operation, loads and stores, takes a vmcnt slot and the
    slots retire in order); count_stores=False counts loads only -- the stricter reading under which k_blur_stream's vmcnt(12) is exact."""
    CAP = 64   # vmcnt holds 6 bits
    n_loads = sum(1 for op, _ in ins if _LOAD.match(op))
    n_waits = sum(1 for op, a in ins if op == "s_waitcnt" and "vmcnt" in a)
    state = {0: {}}
    work = [0]
    viol = {}

    def flow(dst, st):
        if dst >= len(ins):
            return
        old = state.get(dst)
        if old is None:
            state[dst] = dict(st)
            work.append(dst)
            return
        changed = False
        for r, n in st.items():
            if r not in old or old[r] > n:
                old[r] = n
                changed = True
        if changed:
            work.append(dst)

    while work:
        pc = work.pop()
        st = dict(state[pc])
        op, args = ins[pc]
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", args)
            k = None
            if m:
                k = int(m.group(1))
            elif re.fullmatch(r"(0x[0-9a-f]+|\d+)", args):   # raw immediate: vmcnt = bits 3:0 | bits 15:14 << 4 (gfx9)
                v = int(args, 0)
                k = (v & 0xf) | (((v >> 14) & 3) << 4)
            if k is not None:
                st = {r: n for r, n in st.items() if n < k}
            flow(pc + 1, st)
            continue
        if op in ("s_endpgm", "s_trap"):
            continue
        if op == "s_swappc_b64":   # call of a non-inlined device function: the AMDGPU calling convention makes the callee start with
            flow(pc + 1, {})       # s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0) (checked for every such function by callees_wait_at_entry())
            continue
        if op == "s_setpc_b64":
            raise RuntimeError("indirect jump inside a kernel: not analysable")
        used = regs_of(args)
        if _LOAD.match(op):
            parts = [p.strip() for p in args.split(",")]
            returns = not ("atomic" in op and "glc" not in args and "sc0" not in args)
            dest = regs_of(parts[0]) if returns else set()
            # address / data operands must not be outstanding; a destination that is itself still outstanding from an EARLIER load is legal
            # (loads return in order: the later value lands last)
            for r in regs_of(",".join(parts[1:] if returns else parts)) & set(st):
                viol.setdefault((pc, r), (pc, op, args, r))
            st = {r: min(n + 1, CAP) for r, n in st.items()}
            for r in dest:
                st[r] = 0
            flow(pc + 1, st)
            continue
        for r in used & set(st):
            viol.setdefault((pc, r), (pc, op, args, r))
        if count_stores and _STORE.match(op):
            st = {r: min(n + 1, CAP) for r, n in st.items()}
        if op == "s_branch":
            flow(labels[args.split()[0]], st)
            continue
        if op.startswith("s_cbranch"):
            flow(labels[args.split()[-1].strip()], st)
        flow(pc + 1, st)
    return sorted(viol.values()), n_loads, n_waits


def callees_wait_at_entry(asm):
    """every non-kernel function of the unit starts with a full s_waitcnt (the convention check_vmem relies on at s_swappc_b64)"""
    is_kernel = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm, re.M))
    bad = []
    lines = asm.split("\n")
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if not m or m.group(1) in is_kernel:
            continue
        first = next((l for l in lines[i + 1:i + 40] if re.match(r"^\t[a-z]", l)), "")
        if not re.match(r"^\ts_waitcnt vmcnt\(0\)", first):
            bad.append((m.group(1), first.strip()))
    return bad


def report(extra_flags=()):
    import collections
    res = collections.OrderedDict()
    for unit in UNITS:
        asm = compile_unit(unit, extra_flags)
        bad = callees_wait_at_entry(asm)
        if bad:
            raise RuntimeError(f"device functions without a full wait at entry: {bad}")
        for name, (ins, labels, meta) in split_kernels(asm).items():
            v, nl, nw = check_vmem(ins, labels, count_stores="k_blur_stream" not in name)
            res[name] = {"unit": unit, "violations": v, "loads": nl, "vmcnt_waits": nw, "instructions": len(ins), **meta}
    return res


def main():
    res = report()
    bad = 0
    for name, r in res.items():
        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = re.sub(r"\(.*", "", d).replace("orbx::", "").replace("void ", "")
        print(f"{short:34s} instr {r['instructions']:5d}  vmem loads {r['loads']:3d}  vmcnt waits {r['vmcnt_waits']:3d}  vgprs {r.get('NumVgprs', 0):3d}"
              f"  waves/simd {r.get('Occupancy', 0)}  lds {r.get('LDSByteSize', 0):6d}  scratch {r.get('ScratchSize', 0):3d}  violations {len(r['violations'])}")
        for v in r["violations"][:5]:
            print("    ", v)
        bad += len(r["violations"])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
