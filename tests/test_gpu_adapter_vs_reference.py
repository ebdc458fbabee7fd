"""The drop-in boundary, compiled and run: the C++ adapters with the REFERENCE's own signatures against the compiled reference.

* matchers: oracle/ref_matcher_shim.cc -- the test shim that drives the reference's ORBmatcher.cc through stand-in Frame / KeyFrame /
  MapPoint objects -- is compiled a second time against orb_slam3_amd/cpp/ORBmatcher.h + ORBmatcher_slam.inl (same class name, same
  member signatures, include/ORBmatcher.h:47-87) -> oracle/_ref/libmatcher_adapter.so.  With ORBX_MATCHER_BACKEND=adapter the test
  modules written against the compiled reference run UNCHANGED against adapter -> C ABI -> HIP kernels; every result must equal the
  oracle's and the committed outputs of the compiled reference (tests/golden/matchers*_ref.npz).
* extractor: orb_slam3_amd/cpp/ORBextractor.h built with -DORBX_WITH_OPENCV against the OpenCV stand-in (oracle/ocv_shim): the
  reference signature operator()(InputArray, InputArray, vector<KeyPoint>&, OutputArray, vector<int>&) and the lazily fetched
  mvImagePyramid, end to end on the GPU.
"""
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_reference_signature_matchers_run_the_reference_test_modules():
    lib = ROOT / "oracle" / "_ref" / "libmatcher_adapter.so"
    if not lib.exists():
        pytest.skip("oracle/_ref/libmatcher_adapter.so not built (needs /root/reference at build time; it travels with gpurun)")
    env = dict(os.environ, ORBX_MATCHER_BACKEND="adapter")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_oracle_matchers_vs_reference.py", "tests/test_oracle_matchers_small_cases.py",
                        "-q", "-x", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 17, tail


def test_reference_signature_extractor_with_opencv_types(tmp_path):
    from orb_slam3_amd import _lib, synth
    import orb_slam3_amd as osa
    exe = tmp_path / "adapter_ocv_demo"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-DORBX_WITH_OPENCV", "-I", str(ROOT / "oracle" / "ocv_shim"), str(ROOT / "tests/cpp/adapter_ocv_demo.cpp"),
                        "-o", str(exe), str(_lib.LIB_PATH), str(ROOT / "oracle" / "liborb_oracle.so"), "-Wl,-rpath," + str(_lib.LIB_PATH.parent),
                        "-Wl,-rpath," + str(ROOT / "oracle"), "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    img = synth.frame_from_canvas(synth.make_canvas(1), 3, 752, 480, 1003)
    pgm = tmp_path / "f.pgm"
    with open(pgm, "wb") as f:
        f.write(b"P5\n752 480\n255\n" + img.tobytes())
    env = dict(os.environ)
    import torch
    env["LD_LIBRARY_PATH"] = str(Path(torch.__file__).parent / "lib") + ":" + env.get("LD_LIBRARY_PATH", "")   # the HIP runtime the tests use
    out = tmp_path / "out.bin"
    r = subprocess.run([str(exe), str(pgm), str(out)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    ex = osa.ORBextractor(1000, 1.2, 8, 20, 7)
    mono, kps, desc = ex(img, None, (0, 1000))
    raw = np.fromfile(out, np.uint8)
    n = int(raw[:8].view(np.int32)[0])
    assert int(raw[:8].view(np.int32)[1]) == mono and n == len(kps)
    off = 8
    assert raw[off:off + 28 * n].tobytes() == kps.tobytes()
    off += 28 * n
    assert np.array_equal(raw[off:off + 32 * n].reshape(n, 32), desc)
    off += 32 * n
    lvl = ex.get_level(3)[19:-19, 19:-19]
    assert np.array_equal(raw[off:off + lvl.size].reshape(lvl.shape), lvl)   # mvImagePyramid[3] fetched lazily
    assert "levels touched 1" in r.stdout


@pytest.mark.parametrize("case,n_left,n_right,mono_left,mono_right", [(0, 600, 580, 350, 330), (1, 300, 40, 0, 39), (2, 50, 60, 50, 10),
                                                                      (3, 64, 64, 10, 64), (4, 200, 220, 199, 0), (5, 2500, 2400, 700, 650)])
def test_compute_stereo_fisheye_matches(case, n_left, n_right, mono_left, mono_right):
    """Frame::ComputeStereoFishEyeMatches (Frame.cc:1126-1166) through the product: the C++ adapter member (kNN-2 on the GPU, ratio test and
    bookkeeping in the adapter, the camera's TriangulateMatches called back) and its Python mirror, against the oracle and against the
    reference's own text (oracle/_ref/libframe_ref.so live, or its committed outputs in tests/golden/frame_ref.npz)."""
    sys.path.insert(0, str(ROOT / "tests"))
    from test_oracle_frame_vs_reference import fisheye_case, fisheye_triangulate, _P
    from oracle import oracle_binding as ob
    from oracle import ref_binding as rb
    import orb_slam3_amd as osa
    kl, dl, kr, dr = fisheye_case(40 + case, n_left, n_right, mono_left, mono_right)
    sigma2 = np.float32(1.2) ** (2 * np.arange(8, dtype=np.float32))
    tri = fisheye_triangulate(kl, kr)
    on, ond, *o = ob.stereo_fisheye_matches(kl, dl, mono_left, kr, dr, mono_right, sigma2, tri)
    pn, pnd, *pm = osa.ORBmatcher().compute_stereo_fisheye_matches(kl, dl, mono_left, kr, dr, mono_right, sigma2, tri)
    assert (pn, pnd) == (on, ond)
    for a, b in zip(pm, o):
        assert a.tobytes() == b.tobytes()
    if case < 5:   # the pinned cases of tests/test_oracle_frame_vs_reference.py
        _P.pin(f"fisheye_stereo/{case}", (pn, *pm), lambda: rb.ref_stereo_fisheye_matches(kl, dl, mono_left, kr, dr, mono_right, sigma2, tri))
    if (ROOT / "oracle" / "_ref" / "libmatcher_adapter.so").exists():
        an, *am = rb.adapter_stereo_fisheye_matches(kl, dl, mono_left, kr, dr, mono_right, sigma2, tri)
        assert an == on
        for a, b in zip(am, o):
            assert a.tobytes() == b.tobytes()
    if case in (0, 5):
        assert on > 50
