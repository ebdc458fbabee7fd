"""Properties of the gfx950 ISA that no functional test can see (CPU box: hipcc --cuda-device-only -S, no GPU).

k_blur_stream issues its row loads as inline assembly and retires them with a hand-counted `s_waitcnt vmcnt(12)`; the compiler's own
wait insertion does not know about them.  A compiler upgrade or a source edit that adds a copy of an in-flight register, reorders a
VMEM operation or changes the number of loads per row slot would read stale pixels without failing to build.  tools/isa_vmem_check.py
walks the control-flow graph of the emitted code and proves, for every path, that no instruction touches a load's destination before a
wait has retired it.  The same walk over every other kernel checks the compiler's own schedule (and the walker itself: it has to agree
with LLVM on some 40 kernels).  The second half pins the register / LDS / scratch budgets the occupancy figures in DESIGN.md rely on.
"""
import re
import shutil
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="hipcc not installed")


@pytest.fixture(scope="module")
def isa():
    import isa_vmem_check as chk
    units = {u: chk.compile_unit(u) for u in chk.UNITS}
    kernels = {}
    for u, asm in units.items():
        for name, (ins, labels, meta) in chk.split_kernels(asm).items():
            kernels[name] = (ins, labels, meta)
    return chk, units, kernels


def _find(kernels, needle):
    hits = [k for k in kernels if needle in k]
    assert hits, f"no kernel matching {needle}"
    return hits


def test_blur_stream_never_touches_an_inflight_row(isa):
    chk, _, kernels = isa
    for name in _find(kernels, "k_blur_stream"):
        ins, labels, _ = kernels[name]
        viol, n_loads, n_waits = chk.check_vmem(ins, labels, count_stores=False)   # loads only: the strict reading, vmcnt(12) is exact
        assert not viol, (name, viol[:4])
        viol, _, _ = chk.check_vmem(ins, labels, count_stores=True)
        assert not viol, (name, viol[:4])
        # the hand count: 6 row slots x 2 loads in the prologue + 12 slots x 2 loads in the two unrolled groups; every slot waits once
        assert n_loads == 36, (name, n_loads)
        assert sum(1 for op, a in ins if op == "s_waitcnt" and "vmcnt(12)" in a) == 12, name
        # exactly two loads between consecutive row waits (a third would make vmcnt(12) one row short)
        since = None
        for op, a in ins:
            if op == "global_load_dword" and since is not None:
                since += 1
            if op == "s_waitcnt" and "vmcnt(12)" in a:
                assert since in (None, 2), (name, since)
                since = 0


def test_the_walker_detects_a_wait_that_is_one_short(isa):
    """the same code with every vmcnt(12) relaxed to vmcnt(13), and with one load's destination read right after issue: both must be reported"""
    chk, _, kernels = isa
    name = _find(kernels, "k_blur_stream")[0]
    ins, labels, _ = kernels[name]
    relaxed = [(op, a.replace("vmcnt(12)", "vmcnt(13)")) for op, a in ins]
    viol, _, _ = chk.check_vmem(relaxed, labels, count_stores=False)
    assert viol, "a wait one row short went unnoticed"
    i = max(j for j, (op, _) in enumerate(ins) if op == "global_load_dword")
    dst = ins[i][1].split(",")[0].strip()
    touched = ins[:i + 1] + [("v_mov_b32_e32", f"{dst}, {dst}")] + ins[i + 1:]
    shifted = {l: (p if p <= i + 1 else p + 1) for l, p in labels.items()}
    viol, _, _ = chk.check_vmem(touched, shifted, count_stores=False)
    assert any(v[3] == dst for v in viol), "a read of an in-flight destination went unnoticed"


def test_every_kernel_respects_its_outstanding_loads(isa):
    chk, units, kernels = isa
    for u, asm in units.items():
        assert not chk.callees_wait_at_entry(asm), u
    assert len(kernels) >= 35
    for name, (ins, labels, _) in kernels.items():
        viol, _, _ = chk.check_vmem(ins, labels, count_stores=True)
        assert not viol, (name, viol[:4])


# kernel -> (max VGPRs, min waves per SIMD, max static LDS bytes, max scratch bytes).  The occupancy figures of DESIGN.md section 4:
#   k_describe_fused  5 workgroups of 256 threads per CU: <= 96 VGPRs and <= 32 KB of LDS each
#   k_fast_strip<4>   7 workgroups per CU (LDS is dynamic: fast_strip_lds_bytes, checked on the host side): <= 72 VGPRs
#   k_octree_par_t    5 workgroups per CU: <= 96 VGPRs (its 32 bytes of scratch are the frame of the cold lane-0 std::sort fallback call)
BUDGETS = {
    "k_describe_fusedILb0": (96, 5, 32768, 0),
    "k_describe_fusedILb1": (96, 5, 32768, 0),
    "k_describeEPK": (72, 7, 23040, 0),
    "k_fast_stripILi4": (72, 7, 0, 0),
    "k_fast_wave_listILi48": (80, 6, 0, 0),
    "k_octree_par_tILi1792": (96, 5, 64, 32),
    "k_blur_streamILb0": (72, 7, 0, 0),
    "k_blur_streamILb1": (72, 7, 0, 0),
    "k_pyr_resize_marchILi8": (64, 8, 0, 0),
    "k_pyr_stream": (72, 7, 0, 0),          # two 576-thread workgroups per CU (18 waves): <= 7 waves per SIMD are needed, no spill of the task registers
    "k_pyr_base": (32, 8, 0, 0),
    "k_window_best2_tILi8": (64, 8, 0, 0),
    "k_greedy_resolve_tILb0": (104, 4, 128, 0),  # the batched pipeline's form: ONE wave per frame pair beside the next batch's extraction (round 6: lists of 6 entries, 87 -> 98 VGPRs)
    "k_greedy_resolve_tILb1": (144, 3, 128, 0),  # forced single small calls (ORBX_RESOLVE_WAVES=1): + the grid-less re-scan with four features per lane in flight
    "k_resolve_wide_tILi4ELb0": (96, 5, 3072, 0),   # single calls, batched map-point search: 4 waves per problem; static LDS must leave the 160 064 dynamic bytes of the largest frame
    "k_finalize": (32, 8, 64, 0),
}


def test_resource_budgets(isa):
    _, _, kernels = isa
    for needle, (vg, occ, lds, scratch) in BUDGETS.items():
        for name in _find(kernels, needle):
            m = kernels[name][2]
            assert m["NumVgprs"] <= vg, (name, m)
            assert m["Occupancy"] >= occ, (name, m)
            assert m["LDSByteSize"] <= lds, (name, m)
            assert m["ScratchSize"] <= scratch, (name, m)
    # no kernel spills: scratch only where a cold non-inlined call needs a frame
    for name, (_, _, m) in kernels.items():
        if not re.search(r"k_octree|k_debug_sort|k_replay_bowE|k_tri_kb8|k_debug_kb8_gate", name):   # k_replay_bow: the cold big-node path is a non-inlined call; k_tri_kb8: the gate (kb8_gate) is one
            assert m["ScratchSize"] == 0, (name, m)
