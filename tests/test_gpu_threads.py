"""Re-entrancy of the C ABI across host threads ("ctx per host thread => re-entrant").

ORB-SLAM3 calls this path from three threads at once: Tracking (extraction, SearchByProjection x2: Tracking.cc:2856-2894, 3390-3413),
LocalMapping (SearchForTriangulation, Fuse, ComputeDistinctiveDescriptors: LocalMapping.cc:412, 611-720) and LoopClosing (SearchByBoW(KF, KF),
SearchByProjection with a Sim3: LoopClosing.cc:591, 755-777); the stereo Frame constructor runs the left and right extractors on two
std::threads (Frame.cc:122-125).  Here every thread owns its contexts (extractor / matcher objects), all of them hammer one GPU through
liborbx.so concurrently (ctypes releases the GIL for the duration of a call), and EVERY result of EVERY iteration must equal the oracle's.
"""
import os
import subprocess
import threading
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
ITER = int(os.environ.get("ORBX_THREAD_TEST_ITER", "200"))


def _noisy_copy(rng, d, p):
    flip = (rng.random((len(d), 256)) < p)
    return d ^ np.packbits(flip, axis=1, bitorder="little")


def _bow_nodes(rng, k_a, k_b, n_nodes=100, noise=0.15):
    def node(k):
        return ((np.floor(k["x"] / 60).astype(np.int64) * 7 + np.floor(k["y"] / 60).astype(np.int64) * 13 + k["octave"] * 31) % n_nodes)
    na, nb = node(k_a), node(k_b)
    flip = rng.random(len(nb)) < noise
    nb[flip] = rng.integers(0, n_nodes, flip.sum())
    return na, nb


def test_three_slam_threads_share_one_gpu(oracle, canvas1):
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    W, H = 752, 480
    imgs = [synth.frame_from_canvas(canvas1, t, W, H, 1000 + t) for t in range(2)]
    oex = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
    ofr = [oex.extract(im, lap=(0, 1000)) for im in imgs]            # (mono, kps, desc) per frame
    (_, k0, d0), (_, k1, d1) = ofr
    sf = oex.tables()["scale"]
    isg = np.float32(1.0) / (sf * sf)
    g0, g1 = oracle.OracleGrid(k0, 0.0, float(W), 0.0, float(H)), oracle.OracleGrid(k1, 0.0, float(W), 0.0, float(H))
    rng = np.random.default_rng(404)

    # ---- thread A (Tracking): extraction + M1 + M2 ----
    n_mp = 4000
    idx = rng.integers(0, len(k0), n_mp)
    mp = dict(proj_x=(k0["x"][idx] - 2.0 + rng.normal(0, 2, n_mp)).astype(np.float32), proj_y=(k0["y"][idx] - 1.0 + rng.normal(0, 2, n_mp)).astype(np.float32),
              proj_xr=np.zeros(n_mp, np.float32), level=k0["octave"][idx].astype(np.int32), view_cos=rng.uniform(0.9, 1.0, n_mp).astype(np.float32),
              desc=_noisy_copy(rng, d0[idx], 0.04), in_view=(rng.random(n_mp) < 0.95).astype(np.uint8), has_obs=(rng.random(n_mp) < 0.97).astype(np.uint8))
    occ1 = (rng.random(len(k1)) < 0.1).astype(np.uint8)
    want_m1 = oracle.search_by_projection_mappoints(g1, d1, sf, mp, 3.0, 0.8, None, occ1)
    q2 = dict(u=k0["x"] - 2.0, v=k0["y"] - 1.0, ur=np.zeros(len(k0), np.float32), octave=k0["octave"], angle=k0["angle"], desc=d0,
              has_obs=(rng.random(len(k0)) < 0.9).astype(np.uint8))
    want_m2 = oracle.search_by_projection_frame(g1, d1, sf, q2, 15.0, 0, True, None, occ1)

    def tracking():
        ex = osa.ORBextractor(1000, 1.2, 8, 20, 7)
        m1, m2 = osa.ORBmatcher(0.8, True), osa.ORBmatcher(0.9, True)
        F1 = osa.FrameView(k1, d1, 0.0, float(W), 0.0, float(H), sf)
        for it in range(ITER):
            mono, kps, desc = ex(imgs[it & 1], None, (0, 1000))
            omono, okps, odesc = ofr[it & 1]
            assert mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc), ("extract", it)
            n, fm = m1.SearchByProjection(F1, mp, 3.0, occ1)
            assert n == want_m1[0] and np.array_equal(fm, want_m1[1]), ("M1", it)
            n, cm = m2.SearchByProjectionFrame(F1, q2, 15.0, 0, occ1)
            assert n == want_m2[0] and np.array_equal(cm, want_m2[1]), ("M2", it)

    # ---- thread B (LocalMapping): M7 + Fuse + distinctive descriptors ----
    na, nb = _bow_nodes(rng, k0, k1, n_nodes=60)
    fva, fvb = osa.FeatureVector.from_node_of_feature(na), osa.FeatureVector.from_node_of_feature(nb)
    skip0, skip1 = (rng.random(len(k0)) < 0.4).astype(np.uint8), (rng.random(len(k1)) < 0.4).astype(np.uint8)
    want_m7 = oracle.search_for_triangulation(d0, k0["angle"], skip0, fva, d1, k1["angle"], skip1, fvb, True, None)
    lvl = k0["octave"]
    qf = dict(u=k0["x"] - 2.0 + rng.normal(0, 1.2, len(k0)).astype(np.float32), v=k0["y"] - 1.0 + rng.normal(0, 1.2, len(k0)).astype(np.float32),
              ur=(k0["x"] - 20.0).astype(np.float32), r=(np.float32(3.0) * sf[lvl]).astype(np.float32), level=lvl, desc=d0)
    want_fuse = oracle.fuse_search(g1, d1, None, isg, qf, fma=True)
    sizes = list(rng.integers(1, 40, 200)) + [0, 1, 64, 65, 130]
    set_ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    base = rng.integers(0, 256, (len(sizes), 32), dtype=np.uint8)
    dd = np.concatenate([_noisy_copy(rng, np.repeat(base[i:i + 1], n, axis=0), 0.08) for i, n in enumerate(sizes) if n > 0])
    want_dist = oracle.distinctive_descriptors(dd, set_ptr)

    def local_mapping():
        m = osa.ORBmatcher(0.6, True)
        F1 = osa.FrameView(k1, d1, 0.0, float(W), 0.0, float(H), sf, None)
        for it in range(ITER):
            n, m12 = m.SearchForTriangulation(d0, k0["angle"], skip0, fva, d1, k1["angle"], skip1, fvb, None)
            assert n == want_m7[0] and np.array_equal(m12, want_m7[1]), ("M7", it)
            bi, bd = m.FuseSearch(F1, qf, isg, False)
            assert np.array_equal(bi, want_fuse[0]) and np.array_equal(bd, want_fuse[1]), ("Fuse", it)
            got = m.DistinctiveDescriptors(dd, set_ptr)
            assert np.array_equal(got, want_dist), ("distinctive", it)

    # ---- thread C (LoopClosing): SearchByBoW(KF, KF) + M4 ----
    na2, nb2 = _bow_nodes(rng, k0, k1)
    fva2, fvb2 = osa.FeatureVector.from_node_of_feature(na2), osa.FeatureVector.from_node_of_feature(nb2)
    valid0, valid1 = (rng.random(len(k0)) < 0.7).astype(np.uint8), (rng.random(len(k1)) < 0.8).astype(np.uint8)
    want_bow = oracle.search_by_bow_keyframes(d0, k0["angle"], valid0, fva2, d1, k1["angle"], valid1, fvb2, 0.75, True)
    q4 = dict(x=k0["x"] - 2.0 + rng.normal(0, 1.0, len(k0)).astype(np.float32), y=k0["y"] - 1.0, angle=k0["angle"], desc=d0,
              r=(np.float32(8) * sf[lvl]).astype(np.float32), min_level=lvl - 1, max_level=lvl)
    want_m4 = oracle.search_by_projection_window(g1, d1, q4, 75.0, False, True, occ1)

    def loop_closing():
        mb, m4 = osa.ORBmatcher(0.75, True), osa.ORBmatcher(0.75, True)
        F1 = osa.FrameView(k1, d1, 0.0, float(W), 0.0, float(H), sf)
        for it in range(ITER):
            n, m12 = mb.SearchByBoWKeyFrames(d0, k0["angle"], valid0, fva2, d1, k1["angle"], valid1, fvb2)
            assert n == want_bow[0] and np.array_equal(m12, want_bow[1]), ("BoW", it)
            n, mt = m4.SearchByProjectionWindow(F1, q4, 75.0, False, occ1)
            assert n == want_m4[0] and np.array_equal(mt, want_m4[1]), ("M4", it)

    assert want_m1[0] > 100 and want_m2[0] > 100 and want_m7[0] > 30 and want_bow[0] > 30 and want_m4[0] > 100
    if os.environ.get("ORBX_TEST_EMULATOR"):   # the SIMT emulator is single-threaded (fibers + one LDS arena): the same calls, one thread, one iteration
        global ITER
        ITER = 1
        for fn in (tracking, local_mapping, loop_closing):
            fn()
        return
    start = threading.Barrier(3)
    errors = []

    def guarded(fn):
        def run():
            try:
                start.wait(timeout=120)
                fn()
            except BaseException as e:   # noqa: BLE001 -- reported below, with the thread's name
                errors.append((fn.__name__, repr(e)))
        return run

    threads = [threading.Thread(target=guarded(f), name=f.__name__) for f in (tracking, local_mapping, loop_closing)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.skipif(bool(os.environ.get("ORBX_TEST_EMULATOR")), reason="the SIMT emulator is single-threaded")
def test_left_and_right_extractors_on_two_std_threads(tmp_path):
    """Frame.cc:122-125: `thread threadLeft(&Frame::ExtractORB, this, 0, imLeft, ...); thread threadRight(...); join; join` -- the C++ adapter
    classes on two std::threads, 50 stereo frames, every left / right result equal to the same extractor's single-threaded result."""
    from orb_slam3_amd import _lib, synth
    import orb_slam3_amd as osa
    exe = tmp_path / "stereo_threads_demo"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", str(ROOT / "tests/cpp/stereo_threads_demo.cpp"), "-o", str(exe), str(_lib.LIB_PATH),
                        "-Wl,-rpath," + str(_lib.LIB_PATH.parent), "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    w, h = 1241, 376
    canvas = synth.make_canvas(3, size=2600, n_shapes=4000)
    pairs = [synth.make_stereo_pair(3, t, w, h, canvas) for t in range(2)]
    raw = tmp_path / "pairs.bin"
    with open(raw, "wb") as f:
        for L, R in pairs:
            f.write(L.tobytes())
            f.write(R.tobytes())
    env = dict(os.environ)
    import torch
    env["LD_LIBRARY_PATH"] = str(Path(torch.__file__).parent / "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    out = tmp_path / "out.bin"
    r = subprocess.run([str(exe), str(raw), str(w), str(h), "2", "50", str(out)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches 0" in r.stdout, r.stdout
    # and what the threads produced is what the Python mirror (and through the other tests, the oracle) produces
    ex = osa.ORBextractor(2000, 1.2, 8, 20, 7)
    data = np.fromfile(out, np.uint8)
    off = 0
    for L, R in pairs:
        for img in (L, R):
            n = int(data[off:off + 4].view(np.int32)[0])
            off += 4
            _, kps, desc = ex(img, None, (0, 0))
            assert n == len(kps) and data[off:off + 28 * n].tobytes() == kps.tobytes()
            off += 28 * n
            assert np.array_equal(data[off:off + 32 * n].reshape(n, 32), desc)
            off += 32 * n


def test_one_process_one_worker_thread_per_gpu(tmp_path):
    """SURVEY.md 8(e)'s partitioning in the shape of a C++ SLAM process (tests/cpp/multi_gpu_demo.cpp): N std::threads, thread s on GPU s mod G with
    its own extractor, streams and pinned staging ring, extract + frame-to-frame match + asynchronous download per step.  Two workers on the box's
    one GPU, four steps each: every step of a worker equals its first, and the first equals the CPU oracle (keypoints, descriptors, match vectors)."""
    from orb_slam3_amd import _lib, synth
    from oracle import oracle_binding as ob
    exe = tmp_path / "multi_gpu_demo"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(ROOT / "tests/cpp/multi_gpu_demo.cpp"), "-o", str(exe),
                        str(_lib.LIB_PATH), "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + str(_lib.LIB_PATH.parent), "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    w, h, nf, nt = 752, 480, 16, 2
    seqs = []
    for s in range(nt):
        canvas = synth.make_canvas(10 + s)
        seqs.append(np.stack([synth.frame_from_canvas(canvas, t, w, h, 1000 * (10 + s) + t) for t in range(nf)]))
    raw = tmp_path / "frames.bin"
    with open(raw, "wb") as f:
        for q in seqs:
            f.write(q.tobytes())
    env = dict(os.environ)
    import torch
    env["LD_LIBRARY_PATH"] = str(Path(torch.__file__).parent / "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    out = tmp_path / "out.bin"
    r = subprocess.run([str(exe), str(raw), str(w), str(h), str(nf), str(nt), "4", str(out)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches 0" in r.stdout, r.stdout
    data = np.fromfile(out, np.uint8)
    off = 0
    oex = ob.OracleExtractor(1000, 1.2, 8, 20, 7, flags=ob.FLAG_DESC_FMA)
    sf = oex.tables()["scale"]
    for s in range(nt):
        cap = int(data[off:off + 4].view(np.int32)[0]); off += 4
        prev = None
        for t in range(nf):
            n, mono = (int(v) for v in data[off:off + 8].view(np.int32)); off += 8
            omono, okps, odesc = oex.extract(seqs[s][t], lap=(0, 1000))
            assert n == len(okps) and mono == omono, (s, t, n, len(okps))
            assert data[off:off + 28 * n].tobytes() == okps.tobytes(), (s, t); off += 28 * n
            assert np.array_equal(data[off:off + 32 * n].reshape(n, 32), odesc), (s, t); off += 32 * n
            match = data[off:off + 4 * n].view(np.int32).copy(); off += 4 * n
            if prev is not None:   # SearchByProjection(frame t, frame t - 1), th 15, the bench's constant-motion prediction (-2, -1)
                pk, pd = prev
                q = dict(u=pk["x"] - 2.0, v=pk["y"] - 1.0, ur=np.zeros(len(pk), np.float32), octave=pk["octave"], angle=pk["angle"], desc=pd,
                         has_obs=np.ones(len(pk), np.uint8))
                grid = ob.OracleGrid(okps, 0.0, float(w), 0.0, float(h))
                on, ocm = ob.search_by_projection_frame(grid, odesc, sf, q, 15.0, 0, True, None, None)
                assert np.array_equal(match, ocm), (s, t, int((match != ocm).sum()))
            prev = (okps, odesc)
    assert off == len(data)
