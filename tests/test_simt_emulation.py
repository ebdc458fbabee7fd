"""liborbx's own host + device sources compiled for the CPU SIMT emulator (tests/simt/) and compared bit for bit with the oracle.

This exercises the KERNEL LOGIC of the product without a GPU (the `-m gpu` tests remain the parity tests proper: they run the real
code objects on the hardware).  Runs in child processes: the ctypes loader is pointed at the emulated library from test code only."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SIMT = ROOT / "tests" / "simt"
CLANG = Path("/opt/rocm/lib/llvm/bin/clang++")

pytestmark = pytest.mark.skipif(not CLANG.exists(), reason="host clang++ of the ROCm toolchain not found")

PRELUDE = f"""
import sys, numpy as np
sys.path.insert(0, {str(ROOT)!r}); sys.path.insert(0, {str(ROOT / 'tests')!r})
from pathlib import Path
import orb_slam3_amd._lib as _lib
_lib.LIB_PATH = Path({str(SIMT / 'build' / 'liborbx_emul.so')!r})
import orb_slam3_amd as osa
from orb_slam3_amd import synth
from oracle import oracle_binding as ob
"""


@pytest.fixture(scope="module")
def emul_lib():
    r = subprocess.run([sys.executable, str(SIMT / "build.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return SIMT / "build" / "liborbx_emul.so"


def _child(code, env=None, timeout=1500):
    r = subprocess.run([sys.executable, "-c", PRELUDE + code], capture_output=True, text=True, env=dict(os.environ, **(env or {})), timeout=timeout)
    assert r.returncode == 0 and "emulation ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    return r.stdout


STAGEWISE = """
import test_gpu_extractor as tg
img = IMG
n = tg._check_frame(osa.ORBextractor(NF, 1.2, 8, 20, 7), ob.OracleExtractor(NF, 1.2, 8, 20, 7), img, (0, 1000), stagewise=True)
print('emulation ok', n)
"""


@pytest.mark.parametrize("env", [{}, {"SIMT_SHUFFLE": "7"}, {"SIMT_LDS_RANDOM": "3"}, {"ORBX_PYR_CHAIN": "1", "SIMT_SHUFFLE": "5"}])
def test_emulated_extractor_small_image_stagewise(emul_lib, env):
    """Pyramid, blur, FAST candidates, quad-tree output, keypoints and descriptors of the emulated device code == oracle; also with the
    waves of every workgroup resumed in random order, with random garbage in the dynamic LDS, and with the chained pyramid kernel
    (k_pyr_chain: 16 waves per workgroup, levels separated by a workgroup barrier only)."""
    _child(STAGEWISE.replace("IMG", "synth.make_test_image(5, 320, 240)").replace("NF", "500"), env)


def test_emulated_extractor_open_issue_image(emul_lib):
    """The 752x480 frame behind the open 1007-vs-1008 difference seen on the hardware (tests/test_gpu_pipeline.py): the device code's
    logic yields the oracle's 1008 keypoints, stage by stage."""
    out = _child(STAGEWISE.replace("IMG", "synth.frame_from_canvas(synth.make_canvas(11, size=1024, n_shapes=700), 0, 752, 480, 11000)")
                 .replace("NF", "1000"))
    assert "emulation ok 1008" in out


PIPELINE = """
W, H, NF, B = 480, 360, 600, 8
canvases = [synth.make_canvas(10, size=1024, n_shapes=700), synth.make_canvas(11, size=1024, n_shapes=700)]
sets = [np.ascontiguousarray(np.stack([synth.frame_from_canvas(c, t, W, H, 1000 * (10 + i) + t) for t in range(B)])) for i, c in enumerate(canvases)]
ex = osa.ORBextractor(NF, 1.2, 8, 20, 7)
oex = ob.OracleExtractor(NF, 1.2, 8, 20, 7)
sf = oex.tables()["scale"]
cap = ex.output_capacity(W, H)
P = lambda a: a.ctypes.data
for i in range(2):
    hs = dict(kps=np.zeros((B, cap, 28), np.uint8), desc=np.zeros((B, cap, 32), np.uint8), cnt=np.zeros(B, np.int32), mono=np.zeros(B, np.int32),
              match=np.zeros((B, cap), np.int32), nm=np.zeros(B, np.int32))
    (ex.extract_batch_host if i else ex.extract_batch_device)(P(sets[i]), B, W, H, W, W * H, (0, 1000))
    ex.match_consecutive_device(th=15.0, du=-2.0, dv=-1.0, check_orientation=True)
    ex.download_async(P(hs['kps']), P(hs['desc']), P(hs['cnt']), P(hs['mono']), P(hs['match']), P(hs['nm']))
    ex.download_wait()
    prev = None
    for f in range(B):
        mono, k, d = oex.extract(sets[i][f], lap=(0, 1000))
        n = int(hs['cnt'][f])
        assert n == len(k) and int(hs['mono'][f]) == mono and hs['kps'][f, :n].tobytes() == k.tobytes() and np.array_equal(hs['desc'][f, :n], d), (i, f)
        if prev is not None:
            k0, d0 = prev
            q = dict(u=k0["x"] - 2.0, v=k0["y"] - 1.0, ur=np.zeros(len(k0), np.float32), octave=k0["octave"], angle=k0["angle"], desc=d0,
                     has_obs=np.ones(len(k0), np.uint8))
            on, ocm = ob.search_by_projection_frame(ob.OracleGrid(k, 0.0, float(W), 0.0, float(H)), d, sf, q, 15.0, 0, True, None, None)
            assert int(hs['nm'][f]) == on and np.array_equal(hs['match'][f, :len(k)], ocm) and on > 100, (i, f, on)
        prev = (k, d)
print('emulation ok')
"""


@pytest.mark.parametrize("env", [{}, {"ORBX_GRID_BUILD": "2", "ORBX_RESOLVE_RESCAN": "full"}])
def test_emulated_batch_pipeline_with_matcher(emul_lib, env):
    """Two 8-frame batches (from 8 frames on a frame's workgroups are mapped to one XCD: grid (8, blocks, frames / 8)) through
    extract_batch_device / extract_batch_host, the batched frame-to-frame matcher (grid build, window scan, greedy replay with its
    grid re-scan) and the asynchronous download: every frame and every match vector == oracle.  Second run: k_grid_build2 and the
    full-frame re-scan of k_greedy_resolve."""
    _child(PIPELINE, env)


def test_emulated_quadtree_under_wave_shuffle(emul_lib):
    """The quad-tree body alone (tests/simt/octree_emul.cc), both forms, on the oracle's FAST candidates of several frames, every level,
    with the waves of the workgroup resumed in a different random order every scheduler round: always the oracle's DistributeOctTree."""
    code = """
import ctypes as C
L = C.CDLL(str(Path(%r)))
L.simt_octree.restype = C.c_int
pack = lambda c: (c['x'].astype(np.uint32) | (c['y'].astype(np.uint32) << 12) | (c['response'].astype(np.uint32) << 24)).astype(np.uint32)
oex = ob.OracleExtractor(1000, 1.2, 8, 20, 7)
quota = [217, 181, 151, 126, 105, 87, 73, 60]
runs = 0
for seed, size, shapes in ((11, 1024, 700), (10, 2048, 2400), (33, 1536, 4000)):
    c = synth.make_canvas(seed, size=size, n_shapes=shapes)
    for t in range(3):
        oex.extract(synth.frame_from_canvas(c, t, 752, 480, 1000 * seed + t), lap=(0, 1000))
        for l in range(8):
            w, h = oex.level_size(l)
            keys, want = pack(oex.level_candidates(l)), pack(oex.level_keypoints(l))
            for form in (0, 0, 1):   # the 256-thread form twice (different wave orders), then the single-wave chunked form (k_octree_par1)
                out = np.zeros(4096, np.uint32); err = C.c_int(0)
                n = L.simt_octree(form, w, h, quota[l], keys.ctypes.data_as(C.c_void_p), len(keys), out.ctypes.data_as(C.c_void_p), 4096, C.byref(err))
                assert n == len(want) and np.array_equal(out[:n], want) and err.value == 0, (seed, t, l, form, n, len(want), err.value)
                runs += 1
print('emulation ok', runs)
""" % str(SIMT / "build" / "liboctree_emul.so")
    _child(code, {"SIMT_SHUFFLE": "11"})


SWITCHES = [{"ORBX_BLUR_KERNEL": "0"}, {"ORBX_RESIZE_COLS": "1"}, {"ORBX_RESIZE_COLS": "1", "ORBX_RESIZE_PK": "0"}, {"ORBX_PYR_XCD": "0"},
            {"ORBX_PYR_AHEAD": "2"}, {"ORBX_PYR_AHEAD": "1"}, {"ORBX_COPY_AFTER_MATCH": "1"}, {"ORBX_SIDE_STREAMS": "0"}, {"ORBX_BLUR_SIDE": "0"},
            {"ORBX_OCTREE": "seq"}, {"ORBX_FAST_INI": "0"}, {"ORBX_FAST_TPB": "256"}, {"ORBX_BLUR_GROUPS": "3"}, {"ORBX_FAST_INI_QCAP": "48"},
            {"ORBX_FAST_INI_WAVES": "8"}, {"ORBX_PYR_CHAIN": "1"}]


@pytest.mark.skipif(not os.environ.get("ORBX_TEST_EMULATOR_FULL"), reason="opt-in (ORBX_TEST_EMULATOR_FULL=1): about ten minutes")
@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_emulated_switch_matrix(emul_lib, env):
    """Every alternative kernel / scheduling switch of DESIGN.md section 6 through the two-batch pipeline under emulation (their LOGIC:
    slab toggling, grids, the older kernels).  All sixteen passed at the end of round 2."""
    _child(PIPELINE, env)
