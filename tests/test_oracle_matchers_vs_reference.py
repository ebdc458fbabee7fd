"""The oracle's matcher restatements against the REFERENCE ITSELF: /root/reference/src/ORBmatcher.cc compiled where it lies
(oracle/Makefile, _ref/libmatcher_ref.so) against stand-in SLAM types (oracle/mock_slam: Frame / KeyFrame / MapPoint that only
carry data, a camera whose project() is (x, y), identity poses) and the OpenCV shim.  Every loop that runs on the reference
side is the reference's own code: candidate scans, best / second-best bookkeeping, ratio and threshold tests, occupancy
read-after-write, rotation histogram + ComputeThreeMaxima, the BoW merge-walk, vMatchedDistance reassignment, the chi2 gates
of Fuse with the FMA contraction GCC applies to the reference's build flags.  Frame/KeyFrame::GetFeaturesInArea are the
one piece still restated (the mock forwards them to the oracle's grid).

Two layers, like test_oracle_vs_reference.py: a live comparison when oracle/_ref/libmatcher_ref.so is present (it needs
/root/reference at build time; it travels with gpurun), and the committed outputs of that library for the same seeded inputs
(tests/golden/matchers_ref.npz; regenerate with ORBX_WRITE_GOLDEN=1 python -m pytest tests/test_oracle_matchers_vs_reference.py),
which need nothing but the oracle."""
import numpy as np
import pytest

from oracle import oracle_binding as ob
from oracle import ref_binding as rb
from orb_slam3_amd import synth
from orb_slam3_amd.matcher import FeatureVector
from _pin import Pinner

W, H = 752, 480
_P = Pinner("matchers_ref.npz", rb.matcher_available())
_pin = _P.pin


@pytest.fixture(scope="module", autouse=True)
def _write_golden():
    yield
    _P.finish()


def _noisy_copy(rng, d, p):
    flip = rng.random((len(d), 256)) < p
    return d ^ np.packbits(flip, axis=1, bitorder="little")


@pytest.fixture(scope="module")
def frames():
    """Two EuRoC-shaped frames of one canvas (frame t = canvas shifted by (2t, t)), extracted by the oracle."""
    canvas = synth.make_canvas(1)
    out = {}
    for nf, t0, t1 in ((1000, 0, 1), (5000, 0, 4)):
        ex = ob.OracleExtractor(nf, 1.2, 8, 20, 7)
        _, k0, d0 = ex.extract(synth.frame_from_canvas(canvas, t0, W, H, 1000 + t0), lap=(0, 1000))
        _, k1, d1 = ex.extract(synth.frame_from_canvas(canvas, t1, W, H, 1000 + t1), lap=(0, 1000))
        out[nf] = (k0, d0, k1, d1, ex.tables())
    return out


def _views(k, d, sf, u_right=None):
    return ob.OracleGrid(k, 0.0, float(W), 0.0, float(H)), rb.RefFrame(k, d, 0.0, float(W), 0.0, float(H), sf, u_right)


def _depths(rng, n):
    return (rng.integers(8, 320, n) / 8.0).astype(np.float32)   # few mantissa bits: z -+ 1 stays exact


def test_descriptor_distance_and_three_maxima():
    rng = np.random.default_rng(0)
    z, f = np.zeros(32, np.uint8), np.full(32, 255, np.uint8)
    pairs = [(rng.integers(0, 256, 32, dtype=np.uint8), rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(300)] + [(z, f), (f, f)]
    _pin("distance", [ob.descriptor_distance(a, b) for a, b in pairs], lambda: [rb.ref_descriptor_distance(a, b) for a, b in pairs])
    assert ob.descriptor_distance(z, f) == 256 and ob.descriptor_distance(f, f) == 0
    hists = [rng.integers(0, 2 + t % 7, 30) for t in range(3000)] + [np.zeros(30, np.int64)]     # many ties and empty bins
    _pin("three_maxima", [ob.three_maxima(s) for s in hists], lambda: [rb.ref_three_maxima(s) for s in hists])


def test_search_by_projection_mappoints(frames):
    """M1, ORBmatcher.cc:43-213 (mono form): ratio test within a level, stereo window, occupancy read-after-write."""
    k0, d0, k1, d1, tab = frames[1000]
    sf = tab["scale"]
    rng = np.random.default_rng(9)
    n_mp = 3 * len(k0)
    src = rng.integers(0, len(k0), n_mp)
    mp = dict(proj_x=k0["x"][src] - 2.0 + rng.normal(0, 2, n_mp).astype(np.float32),
              proj_y=k0["y"][src] - 1.0 + rng.normal(0, 2, n_mp).astype(np.float32),
              proj_xr=(k0["x"][src] - 2.0 - rng.uniform(0, 30, n_mp)).astype(np.float32), level=k0["octave"][src],
              view_cos=rng.uniform(0.99, 1.0, n_mp).astype(np.float32), desc=_noisy_copy(rng, d0[src], 0.04),
              in_view=(rng.random(n_mp) < 0.95).astype(np.uint8), has_obs=(rng.random(n_mp) < 0.9).astype(np.uint8))
    occ = (rng.random(len(k1)) < 0.1).astype(np.uint8)
    ur = np.where(rng.random(len(k1)) < 0.6, k1["x"] - rng.uniform(0, 30, len(k1)), -1.0).astype(np.float32)
    for th, ratio, u_right in ((1.0, 0.8, None), (3.0, 0.8, None), (3.0, 0.6, ur), (15.0, 0.9, ur)):
        grid, F = _views(k1, d1, sf, u_right)
        on, ofm = ob.search_by_projection_mappoints(grid, d1, sf, mp, th, ratio, u_right, occ)
        _pin(f"m1/{th}/{ratio}/{u_right is not None}", (on, ofm), lambda: rb.ref_search_by_projection_mappoints(F, mp, th, ratio, occ))
        assert on > 100


def test_search_by_projection_mappoints_fisheye_twin(frames):
    """M1 with F.Nleft != -1 (ORBmatcher.cc:43-213 whole): left search and right-camera twin interleaved per map point, partner
    slots written through mvLeftToRightMatch / mvRightToLeftMatch, the twin's radius without the th factor.  Oracle only so far
    (liborbx implements the Nleft == -1 form); this pins the semantics the device twin has to reproduce."""
    k0, d0, k1, d1, tab = frames[1000]
    sf = tab["scale"]
    rng = np.random.default_rng(19)
    kl, dl, kr, dr = k1, d1, k0, d0               # "left" = frame 1, "right" = frame 0 (the same scene shifted by (2, 1))
    nl, nr = len(kl), len(kr)
    desc = np.concatenate([dl, dr])
    l2r = np.full(nl, -1, np.int32)
    r2l = np.full(nr, -1, np.int32)
    pairs = rng.permutation(min(nl, nr))[:300]     # some stereo partners (any injective pairing exercises the partner writes)
    l2r[pairs] = pairs
    r2l[pairs] = pairs
    n_mp = 2500
    src = rng.integers(0, nl, n_mp)
    srcr = rng.integers(0, nr, n_mp)
    mp = dict(in_view=(rng.random(n_mp) < 0.8).astype(np.uint8), proj_x=kl["x"][src] + rng.normal(0, 2, n_mp).astype(np.float32),
              proj_y=kl["y"][src] + rng.normal(0, 2, n_mp).astype(np.float32), level=kl["octave"][src],
              view_cos=rng.choice([0.99, 0.9995], n_mp).astype(np.float32), in_view_r=(rng.random(n_mp) < 0.7).astype(np.uint8),
              proj_xr=kr["x"][srcr] + rng.normal(0, 2, n_mp).astype(np.float32), proj_yr=kr["y"][srcr] + rng.normal(0, 2, n_mp).astype(np.float32),
              level_r=np.where(rng.random(n_mp) < 0.9, kr["octave"][srcr], -1).astype(np.int32),
              view_cos_r=rng.choice([0.99, 0.9995], n_mp).astype(np.float32), desc=_noisy_copy(rng, dl[src], 0.05),
              has_obs=(rng.random(n_mp) < 0.9).astype(np.uint8))
    occ = (rng.random(nl + nr) < 0.1).astype(np.uint8)
    gl, gr = ob.OracleGrid(kl, 0.0, float(W), 0.0, float(H)), ob.OracleGrid(kr, 0.0, float(W), 0.0, float(H))
    bounds = np.array([0.0, W, 0.0, H], np.float32)
    for th, ratio in ((1.0, 0.8), (3.0, 0.8), (5.0, 0.6)):
        on, ofm = ob.search_by_projection_mappoints_fisheye(gl, gr, desc, sf, l2r, r2l, mp, th, ratio, occ)
        _pin(f"m1fisheye/{th}/{ratio}", (on, ofm),
             lambda: rb.ref_search_by_projection_mappoints_fisheye(kl, kr, desc, bounds, sf, l2r, r2l, mp, th, ratio, occ))
        assert on > 200 and (ofm[nl:] >= 0).sum() > 50 and (ofm[:nl] >= 0).sum() > 50


def test_search_by_projection_frame(frames):
    """M2, ORBmatcher.cc:1676-1885: mono / forward / backward level windows, stereo gate u - bf/z, rotation filter."""
    k0, d0, k1, d1, tab = frames[1000]
    sf = tab["scale"]
    rng = np.random.default_rng(5)
    z = _depths(rng, len(k0))
    u = (k0["x"] - 2.0).astype(np.float32)
    q = dict(u=u, v=k0["y"] - 1.0, z=z, ur=(u - np.float32(1.0) / z).astype(np.float32), octave=k0["octave"], angle=k0["angle"],
             desc=d0, has_obs=(rng.random(len(k0)) < 0.9).astype(np.uint8))
    inside = (q["u"] >= 0) & (q["u"] <= W) & (q["v"] >= 0) & (q["v"] <= H)
    q = {k: v[inside] for k, v in q.items()}
    ur = np.where(rng.random(len(k1)) < 0.5, k1["x"] - rng.uniform(0, 20, len(k1)), -1.0).astype(np.float32)
    for mode, ori, th, u_right in ((0, True, 15.0, None), (1, True, 7.0, None), (2, False, 15.0, None), (0, True, 30.0, None),
                                   (1, True, 15.0, ur), (2, True, 15.0, ur)):
        occ = (rng.random(len(k1)) < 0.05).astype(np.uint8)
        grid, F = _views(k1, d1, sf, u_right)
        on, ocm = ob.search_by_projection_frame(grid, d1, sf, q, th, mode, ori, u_right, occ)
        _pin(f"m2/{mode}/{ori}/{th}/{u_right is not None}", (on, ocm), lambda: rb.ref_search_by_projection_frame(F, q, th, mode, ori, occ))
        assert on > 50


def test_search_by_projection_frame_fisheye_twin(frames):
    """M2 with CurrentFrame.Nleft != -1 (ORBmatcher.cc:1676-1885 whole): right-camera twin at the projection into the right camera,
    skipped together with an empty left window, rotation entries from both cameras.  Oracle only so far."""
    k0, d0, k1, d1, tab = frames[1000]
    sf = tab["scale"]
    rng = np.random.default_rng(23)
    kl, dl = k1, d1                               # current left = frame 1
    kr = k1.copy()                                # current right = the same features displaced by a per-feature disparity
    disp = (rng.integers(8, 160, len(kr)) / 8.0).astype(np.float32)
    kr["x"] = kl["x"] + disp
    dr = _noisy_copy(rng, d1, 0.02)
    nl, nr = len(kl), len(kr)
    desc = np.concatenate([dl, dr])
    u = (k0["x"] - 2.0).astype(np.float32)
    v = (k0["y"] - 1.0).astype(np.float32)
    z = (rng.integers(8, 160, len(k0)) / 8.0).astype(np.float32)
    q = dict(u=u, v=v, z=z, xr=(u + z).astype(np.float32), yr=v, octave=k0["octave"], angle=k0["angle"], desc=d0,
             has_obs=(rng.random(len(k0)) < 0.9).astype(np.uint8))
    inside = (u >= 0) & (u <= W) & (v >= 0) & (v <= H)
    q = {k: a[inside] for k, a in q.items()}
    gl, gr = ob.OracleGrid(kl, 0.0, float(W), 0.0, float(H)), ob.OracleGrid(kr, 0.0, float(W), 0.0, float(H))
    bounds = np.array([0.0, W, 0.0, H], np.float32)
    for mode, ori, th in ((0, True, 15.0), (1, True, 7.0), (2, False, 15.0), (0, True, 40.0)):
        occ = (rng.random(nl + nr) < 0.05).astype(np.uint8)
        on, ocm = ob.search_by_projection_frame_fisheye(gl, gr, desc, sf, q, th, mode, ori, occ)
        _pin(f"m2fisheye/{mode}/{ori}/{th}", (on, ocm),
             lambda: rb.ref_search_by_projection_frame_fisheye(kl, kr, desc, bounds, sf, q, th, mode, ori, occ))
        assert (ocm[:nl] >= 0).sum() > 50 and (ocm[nl:] >= 0).sum() > 20


def test_search_by_projection_keyframe_and_sim3(frames):
    """M3, ORBmatcher.cc:1887-2010 (Frame grid, levels [l-1,l+1], ORBdist, rotation filter) and M4, :427-646 both overloads
    (KeyFrame grid + explicit octave gate [l-1,l], TH_LOW * ratioHamming)."""
    k0, d0, k1, d1, tab = frames[1000]
    sf = tab["scale"]
    rng = np.random.default_rng(21)
    lvl = k0["octave"]
    x = (k0["x"] - 2.0 + rng.normal(0, 1.0, len(k0))).astype(np.float32)
    y = (k0["y"] - 1.0).astype(np.float32)
    inside = (x >= 0) & (x < W) & (y >= 0) & (y < H)
    x, y, lvl, ang, dq = x[inside], y[inside], lvl[inside], k0["angle"][inside], _noisy_copy(rng, d0[inside], 0.03)
    occ = (rng.random(len(k1)) < 0.1).astype(np.uint8)
    grid, F = _views(k1, d1, sf)
    for th, orbdist, ori in ((10.0, 100, True), (3.0, 64, True), (10.0, 100, False)):     # Tracking.cc:3726,3740
        q = dict(x=x, y=y, r=(np.float32(th) * sf[lvl]).astype(np.float32), min_level=lvl - 1, max_level=lvl + 1, angle=ang, desc=dq)
        on, om = ob.search_by_projection_window(grid, d1, q, float(orbdist), ori, False, occ)
        _pin(f"m3/{th}/{orbdist}/{ori}", (on, om),
             lambda: rb.ref_search_by_projection_keyframe(F, dict(x=x, y=y, level=lvl, angle=ang, desc=dq), th, orbdist, ori, occ))
        assert on > 100
    # key frame slots without / with bad / with already-found map points drop out of the query list
    skip = rng.choice([0, 0, 0, 1, 2, 3], len(x)).astype(np.uint8)
    keep = np.nonzero(skip == 0)[0]
    q = dict(x=x[keep], y=y[keep], r=(np.float32(10.0) * sf[lvl[keep]]).astype(np.float32), min_level=lvl[keep] - 1,
             max_level=lvl[keep] + 1, angle=ang[keep], desc=dq[keep])
    on, om = ob.search_by_projection_window(grid, d1, q, 100.0, True, False, occ)
    _pin("m3/skips", (on, np.where(om >= 0, keep[np.maximum(om, 0)], -1)),
         lambda: rb.ref_search_by_projection_keyframe(F, dict(x=x, y=y, level=lvl, angle=ang, desc=dq), 10.0, 100, True, occ, skip))
    for th, ratio in ((8, 1.5), (5, 1.0), (3, 1.5)):                                         # LoopClosing.cc:755,777,964
        q = dict(x=x, y=y, r=(np.float32(th) * sf[lvl]).astype(np.float32), min_level=lvl - 1, max_level=lvl, desc=dq)
        on, om = ob.search_by_projection_window(grid, d1, q, 50 * ratio, False, True, occ)
        for variant in (0, 1):
            _pin(f"m4/{th}/{ratio}/{variant}", (on, om),
                 lambda: rb.ref_search_by_projection_sim3(F, dict(x=x, y=y, level=lvl, desc=dq), th, ratio, variant, occ))
        assert on > 100


def test_search_by_projection_keyframe_fisheye_frame(frames):
    """M3 (ORBmatcher.cc:1887-2010) on a fisheye-stereo current frame (Nleft != -1): the reference takes no special path -- its
    GetFeaturesInArea call searches the LEFT camera's grid (bRight defaults to false) and mvKeysUn == mvKeys there -- so the result is the
    mono search over the left features; the right camera's features are neither candidates nor written."""
    k0, d0, k1, d1, tab = frames[1000]
    sf = tab["scale"]
    rng = np.random.default_rng(23)
    lvl = k0["octave"]
    x = (k0["x"] - 2.0 + rng.normal(0, 1.0, len(k0))).astype(np.float32)
    y = (k0["y"] - 1.0).astype(np.float32)
    inside = (x >= 0) & (x < W) & (y >= 0) & (y < H)
    x, y, lvl, ang, dq = x[inside], y[inside], lvl[inside], k0["angle"][inside], _noisy_copy(rng, d0[inside], 0.03)
    nl = len(k1)
    k_right = k1.copy(); k_right["x"] = np.clip(k1["x"] - 7.0, 0, W - 1).astype(np.float32)       # the right camera: shifted copies
    desc_all = np.concatenate([d1, _noisy_copy(rng, d1, 0.01)])                                   # ... with BETTER descriptors than the left ones
    occ_all = (rng.random(2 * nl) < 0.1).astype(np.uint8)
    grid = ob.OracleGrid(k1, 0.0, float(W), 0.0, float(H))
    bounds = np.array([0.0, W, 0.0, H], np.float32)
    for th, orbdist, ori in ((10.0, 100, True), (3.0, 64, False)):
        q = dict(x=x, y=y, r=(np.float32(th) * sf[lvl]).astype(np.float32), min_level=lvl - 1, max_level=lvl + 1, angle=ang, desc=dq)
        on, om = ob.search_by_projection_window(grid, d1, q, float(orbdist), ori, False, occ_all[:nl])
        full = np.concatenate([om, np.full(nl, -1, np.int32)])
        _pin(f"m3_fisheye/{th}/{orbdist}/{ori}", (on, full),
             lambda: rb.ref_search_by_projection_keyframe_fisheye(k1, k_right, desc_all, bounds, sf, dict(x=x, y=y, level=lvl, angle=ang, desc=dq),
                                                                 th, orbdist, ori, occ_all))
        assert on > 100


def _bow_nodes(rng, k_a, k_b, n_nodes=100, noise=0.15):
    def node(k):
        return (np.floor(k["x"] / 60).astype(np.int64) * 7 + np.floor(k["y"] / 60).astype(np.int64) * 13 + k["octave"] * 31) % n_nodes
    na, nb = node(k_a), node(k_b)
    flip = rng.random(len(nb)) < noise
    nb[flip] = rng.integers(0, n_nodes, flip.sum())
    return na, nb


def test_search_by_bow(frames):
    """M5, ORBmatcher.cc:223-425 (KeyFrame -> Frame) and :765-905 (KeyFrame -> KeyFrame) over the reference's own
    DBoW2::FeatureVector (std::map walk with lower_bound)."""
    k0, d0, k1, d1, _ = frames[1000]
    rng = np.random.default_rng(31)
    for n_nodes in (100, 7):
        na, nb = _bow_nodes(rng, k0, k1, n_nodes)
        na[na == 3] = 4   # a node present on one side only: exercises the lower_bound skips
        fva, fvb = FeatureVector.from_node_of_feature(na), FeatureVector.from_node_of_feature(nb)
        valid0 = (rng.random(len(k0)) < 0.7).astype(np.uint8)
        valid1 = (rng.random(len(k1)) < 0.8).astype(np.uint8)
        for ratio, ori in ((0.7, True), (0.9, True), (0.75, False)):
            on, om = ob.search_by_bow_frame(d0, k0["angle"], valid0, fva, d1, k1["angle"], fvb, ratio, ori)
            _pin(f"m5a/{n_nodes}/{ratio}/{ori}", (on, om),
                 lambda: rb.ref_search_by_bow_frame(d0, k0["angle"], valid0, fva, d1, k1["angle"], fvb, ratio, ori))
            assert on > 50
            on, om = ob.search_by_bow_keyframes(d0, k0["angle"], valid0, fva, d1, k1["angle"], valid1, fvb, ratio, ori)
            _pin(f"m5b/{n_nodes}/{ratio}/{ori}", (on, om),
                 lambda: rb.ref_search_by_bow_keyframes(d0, k0["angle"], valid0, fva, d1, k1["angle"], valid1, fvb, ratio, ori))
            assert on > 30


def test_search_by_bow_keyframes_fisheye(frames):
    """M5 (KeyFrame -> KeyFrame) between fisheye-stereo key frames (NLeft != -1, ORBmatcher.cc:800-802 / :820-822): the right camera's
    features (index >= mvKeysUn.size()) are skipped as queries and as candidates -- the oracle's mono form with those features marked as
    having no map point must give the reference's vector."""
    k0, d0, k1, d1, _ = frames[1000]
    rng = np.random.default_rng(41)
    nl0, nl1 = len(k0), len(k1)
    D0 = np.concatenate([d0, _noisy_copy(rng, d0, 0.03)]); D1 = np.concatenate([d1, _noisy_copy(rng, d1, 0.03)])   # right cameras: noisy copies
    A0 = np.concatenate([k0["angle"], k0["angle"]]).astype(np.float32); A1 = np.concatenate([k1["angle"], k1["angle"]]).astype(np.float32)
    for n_nodes in (100, 7):
        na, nb = _bow_nodes(rng, k0, k1, n_nodes)
        fva = FeatureVector.from_node_of_feature(np.concatenate([na, na])); fvb = FeatureVector.from_node_of_feature(np.concatenate([nb, nb]))
        v0 = (rng.random(2 * nl0) < 0.8).astype(np.uint8); v1 = (rng.random(2 * nl1) < 0.8).astype(np.uint8)
        m0 = v0.copy(); m0[nl0:] = 0
        m1 = v1.copy(); m1[nl1:] = 0
        for ratio, ori in ((0.7, True), (0.9, False)):
            on, om = ob.search_by_bow_keyframes(D0, A0, m0, fva, D1, A1, m1, fvb, ratio, ori)
            _pin(f"m5b_fisheye/{n_nodes}/{ratio}/{ori}", (on, om),
                 lambda: rb.ref_search_by_bow_keyframes_fisheye(D0, A0, v0, nl0, fva, D1, A1, v1, nl1, fvb, ratio, ori))
            assert on > 30 and (om[nl0:] == -1).all() and (om < nl1).all()


def test_search_by_bow_frame_fisheye(frames):
    """M5 (KeyFrame -> Frame) with F.Nleft != -1 (ORBmatcher.cc:283-392): per-camera best / second-best over one node list, the right
    match nested inside the left branch with `|| true` for a ratio test.  Oracle only so far."""
    k0, d0, k1, d1, _ = frames[1000]
    rng = np.random.default_rng(37)
    n_left = len(k1)
    f_desc = np.concatenate([d1, _noisy_copy(rng, d1, 0.03)])       # right camera: noisy copies of the left features
    f_angle = np.concatenate([k1["angle"], (k1["angle"] + rng.normal(0, 3, n_left)).astype(np.float32) % 360]).astype(np.float32)
    for n_nodes in (100, 9):
        na, nb = _bow_nodes(rng, k0, k1, n_nodes)
        nbr = nb.copy()
        flip = rng.random(n_left) < 0.1
        nbr[flip] = rng.integers(0, n_nodes, flip.sum())
        fva, fvb = FeatureVector.from_node_of_feature(na), FeatureVector.from_node_of_feature(np.concatenate([nb, nbr]))
        valid0 = (rng.random(len(k0)) < 0.7).astype(np.uint8)
        for ratio, ori in ((0.7, True), (0.9, True), (0.75, False)):
            on, om = ob.search_by_bow_frame_fisheye(d0, k0["angle"], valid0, fva, f_desc, f_angle, n_left, fvb, ratio, ori)
            _pin(f"m5afisheye/{n_nodes}/{ratio}/{ori}", (on, om),
                 lambda: rb.ref_search_by_bow_frame_fisheye(d0, k0["angle"], valid0, fva, f_desc, f_angle, n_left, fvb, ratio, ori))
            assert (om[:n_left] >= 0).sum() > 40 and (om[n_left:] >= 0).sum() > 40
            mono_n, _ = ob.search_by_bow_frame(d0, k0["angle"], valid0, fva, d1, k1["angle"], FeatureVector.from_node_of_feature(nb), ratio, ori)
            assert on > mono_n     # the twin adds right-camera matches


def test_search_for_initialization(frames):
    """M6, ORBmatcher.cc:648-763: level-0 keypoints, 100-px window, vMatchedDistance reassignment, vbPrevMatched update."""
    k0, d0, k1, d1, tab = frames[5000]
    bounds = np.array([0.0, W, 0.0, H], np.float32)
    for ratio, ori, win in ((0.9, True, 100), (0.7, False, 100), (0.9, True, 30)):
        prev_a = np.ascontiguousarray(np.stack([k0["x"], k0["y"]], axis=1).astype(np.float32))
        prev_b = prev_a.copy()
        grid = ob.OracleGrid(k1, 0.0, float(W), 0.0, float(H))
        on, om = ob.search_for_initialization(k0, d0, grid, d1, prev_a, win, ratio, ori)
        _pin(f"m6/{ratio}/{ori}/{win}", (on, om, prev_a),
             lambda: rb.ref_search_for_initialization(k0, d0, k1, d1, bounds, prev_b, win, ratio, ori) + (prev_b,))
        assert on > 200


def contended_init_case(frames):
    """SearchForInitialization inputs with heavy contention for F2's features: level-0 descriptors of BOTH frames are drawn from 40 prototypes with a few
    bits flipped, so that every query sees dozens of candidates within TH_LOW, matches are taken over again and again (vnMatches21 reassignment,
    :713-716), vMatchedDistance skips are frequent and distance ties abound.  Also used by tests/test_gpu_matcher.py."""
    k0, d0, k1, d1, _ = frames[5000]
    rng = np.random.default_rng(88)
    d0c, d1c = d0.copy(), d1.copy()
    proto = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    for k, d in ((k0, d0c), (k1, d1c)):
        lv0 = np.nonzero(k["octave"] == 0)[0]
        pick = rng.integers(0, len(proto), len(lv0))
        flips = (rng.random((len(lv0), 256)) < 0.03)
        d[lv0] = proto[pick] ^ np.packbits(flips, axis=1, bitorder="little")
    return k0, d0c, k1, d1c


def test_search_for_initialization_under_contention(frames):
    """M6 where the loop's state matters (round 6: the device replays the loop over precomputed candidate lists; a list that runs dry is re-scanned)."""
    k0, d0, k1, d1 = contended_init_case(frames)
    bounds = np.array([0.0, W, 0.0, H], np.float32)
    for ratio, ori, win in ((0.9, True, 100), (1.0, False, 60), (0.95, True, 25)):
        prev_a = np.ascontiguousarray(np.stack([k0["x"], k0["y"]], axis=1).astype(np.float32))
        prev_b = prev_a.copy()
        grid = ob.OracleGrid(k1, 0.0, float(W), 0.0, float(H))
        on, om = ob.search_for_initialization(k0, d0, grid, d1, prev_a, win, ratio, ori)
        _pin(f"m6c/{ratio}/{ori}/{win}", (on, om, prev_a),
             lambda: rb.ref_search_for_initialization(k0, d0, k1, d1, bounds, prev_b, win, ratio, ori) + (prev_b,))
        assert on > 20, on


def test_search_for_triangulation(frames):
    """M7, ORBmatcher.cc:907-1146: later equal-distance candidate wins (dist > bestDist skips), vbMatched2 is never set in the
    reference, the epipolar verdict is consulted lazily, rotation filter, bCoarse bypass."""
    k0, d0, k1, d1, _ = frames[1000]
    rng = np.random.default_rng(41)
    na, nb = _bow_nodes(rng, k0, k1, 60)
    fva, fvb = FeatureVector.from_node_of_feature(na), FeatureVector.from_node_of_feature(nb)
    s0 = (rng.random(len(k0)) < 0.3).astype(np.uint8)
    s1 = (rng.random(len(k1)) < 0.3).astype(np.uint8)
    table = (rng.random((len(k0), len(k1))) < 0.7).astype(np.uint8)
    d1b = d1.copy()
    d1b[5] = d1b[9]    # exact duplicates inside KF2
    for ori, tab, coarse in ((True, table, False), (False, table, False), (True, None, False), (True, table, True)):
        pred = None if (tab is None or coarse) else (lambda i, j: tab[i, j])
        on, om = ob.search_for_triangulation(d0, k0["angle"], s0, fva, d1b, k1["angle"], s1, fvb, ori, pred)
        _pin(f"m7/{ori}/{tab is not None}/{coarse}", (on, om),
             lambda: rb.ref_search_for_triangulation(d0, k0["angle"], s0, fva, d1b, k1["angle"], s1, fvb, ori, tab, coarse))
        assert on > 50


def test_search_for_triangulation_pinhole_gates(frames):
    """M7 with the gates the way the device evaluates them (orbx_search_for_triangulation_pinhole): the reference's own
    epipole-distance test (ORBmatcher.cc:1026-1034) runs on the real keypoints, and its epipolarConstrain verdicts come from the
    reference's Pinhole.cpp text (identity intrinsics, R12 = I: F12 = [t12]x exactly)."""
    k0, d0, k1, d1, tab = frames[1000]
    sf, sg = tab["scale"], tab["sigma2"]
    rng = np.random.default_rng(43)
    na, nb = _bow_nodes(rng, k0, k1, 60)
    fva, fvb = FeatureVector.from_node_of_feature(na), FeatureVector.from_node_of_feature(nb)
    s0 = (rng.random(len(k0)) < 0.3).astype(np.uint8)
    s1 = (rng.random(len(k1)) < 0.3).astype(np.uint8)
    t12 = np.array([1.0, 0.02, 0.0005], np.float32)
    F = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]], np.float32)
    ep = (400.0, 240.0)
    ur0 = np.where(rng.random(len(k0)) < 0.2, k0["x"] - 5.0, -1.0).astype(np.float32)
    ur1 = np.where(rng.random(len(k1)) < 0.2, k1["x"] - 5.0, -1.0).astype(np.float32)
    if rb.frame_available():   # verdict table of the reference's epipolarConstrain for every pair (unc = sigma2 of kp2's octave)
        i0, i1 = np.meshgrid(np.arange(len(k0)), np.arange(len(k1)), indexing="ij")
        ok, Fref = rb.ref_epipolar_pinhole([1, 1, 0, 0], [1, 1, 0, 0], np.eye(3, dtype=np.float32), t12, k0["x"][i0.ravel()], k0["y"][i0.ravel()],
                                           k1["x"][i1.ravel()], k1["y"][i1.ravel()], sg[k1["octave"][i1.ravel()]])
        assert np.array_equal(Fref, F)
        table = ok.reshape(len(k0), len(k1))
    for ori, coarse, u0, u1 in ((True, False, None, None), (False, False, ur0, ur1), (True, True, None, ur1)):
        on, om = ob.search_for_triangulation_pinhole(k0, d0, s0, u0, fva, k1, d1, s1, u1, fvb, sf, sg, F, ep, coarse, ori, fma=True)
        _pin(f"m7geo/{ori}/{coarse}/{u0 is not None}", (on, om),
             lambda: rb.ref_search_for_triangulation_geo(k0, d0, s0, u0, fva, k1, d1, s1, u1, fvb, sf, ep, ori, table, coarse))
        assert on > 50
    # the gates do bite: without them more pairs survive
    free, _ = ob.search_for_triangulation(d0, k0["angle"], s0, fva, d1, k1["angle"], s1, fvb, True, None)
    gated, _ = ob.search_for_triangulation_pinhole(k0, d0, s0, None, fva, k1, d1, s1, None, fvb, sf, sg, F, ep, False, True)
    assert gated < free


def _rot(rx, ry, rz):
    """float32 rotation matrix Rz(rz) Ry(ry) Rx(rx)"""
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (Rz @ Ry @ Rx).astype(np.float32)


PINHOLE_CASES = [
    # name, K1, K2, pose2 = Tcw of key frame 2 (key frame 1 sits at the origin), bOnlyStereo, bCoarse, check orientation, stereo flags
    # sideways motion, intrinsics and translation exact in binary: the frames' (2, 1) px shift is this baseline at one depth, many pairs pass
    ("sideways_exact", (512.0, 512.0, 384.0, 256.0), (512.0, 512.0, 384.0, 256.0), (np.eye(3, dtype=np.float32), (-1.0, -0.5, 0.015625)), False, False, True, False),
    # EuRoC cam0 intrinsics, a small rotation, mixed stereo / monocular keypoints
    ("euroc_rotated", (458.654, 457.296, 367.215, 248.375), (458.654, 457.296, 367.215, 248.375), (_rot(0.0004, -0.0007, 0.0011), (-0.11, -0.05, 0.004)), False, False,
     False, True),
    # forward motion: the epipole lies inside image 2, the epipole-distance gate of :1026-1034 fires
    ("forward_epipole_in_image", (458.654, 457.296, 367.215, 248.375), (460.0, 459.0, 370.0, 250.0), (np.eye(3, dtype=np.float32), (0.002, -0.001, -0.4)), False, False, True,
     False),
    ("only_stereo", (512.0, 512.0, 384.0, 256.0), (512.0, 512.0, 384.0, 256.0), (np.eye(3, dtype=np.float32), (-1.0, -0.5, 0.015625)), True, False, True, True),
    ("coarse", (458.654, 457.296, 367.215, 248.375), (458.654, 457.296, 367.215, 248.375), (_rot(0.0004, -0.0007, 0.0011), (-0.11, -0.05, 0.004)), False, True, True, False),
]


def pinhole_case_inputs(frames, case):
    """Inputs of one PINHOLE_CASES entry (also used by tests/test_gpu_matcher.py and bench.py's latency block)."""
    name, K1, K2, pose2, only_stereo, coarse, ori, stereo = case
    k0, d0, k1, d1, tab = frames[1000]
    rng = np.random.default_rng(sum(map(ord, name)))
    na, nb = _bow_nodes(rng, k0, k1, 60)
    fva, fvb = FeatureVector.from_node_of_feature(na), FeatureVector.from_node_of_feature(nb)
    s0 = (rng.random(len(k0)) < 0.3).astype(np.uint8)
    s1 = (rng.random(len(k1)) < 0.3).astype(np.uint8)
    ur0 = np.where(rng.random(len(k0)) < 0.5, k0["x"] - 5.0, -1.0).astype(np.float32) if stereo else None
    ur1 = np.where(rng.random(len(k1)) < 0.5, k1["x"] - 5.0, -1.0).astype(np.float32) if stereo else None
    pose1 = (np.eye(3, dtype=np.float32), np.zeros(3, np.float32))
    pose2 = (np.asarray(pose2[0], np.float32), np.asarray(pose2[1], np.float32))
    return dict(k0=k0, d0=d0, s0=s0, ur0=ur0, fva=fva, k1=k1, d1=d1, s1=s1, ur1=ur1, fvb=fvb, sf=tab["scale"], sg=tab["sigma2"], K1=K1, K2=K2, pose1=pose1, pose2=pose2,
                only_stereo=only_stereo, coarse=coarse, ori=ori)


@pytest.mark.parametrize("case", PINHOLE_CASES, ids=[c[0] for c in PINHOLE_CASES])
def test_search_for_triangulation_between_pinhole_keyframes(frames, case):
    """M7 the way LocalMapping::CreateNewMapPoints calls it (LocalMapping.cc:466): two key frames with Pinhole cameras and poses, nothing
    else.  Reference side: ORBmatcher.cc:907-1146 whole, the epipole from Pinhole::project, Pinhole::epipolarConstrain deriving F12 from
    (K1, K2, R12, t12) for every pair -- all the reference's own text.  The oracle gets the epipole and the F12 that run used and must return
    the same pairs.  With ORBX_MATCHER_BACKEND=adapter the same call goes through the drop-in adapter, which must take its CAM_PINHOLE
    route (the stand-in's host epipolarConstrain aborts in that build): F12 built once, both gates inside k_replay_bow."""
    c = pinhole_case_inputs(frames, case)
    name = case[0]
    live = {}

    def run():
        if "r" not in live:
            live["r"] = rb.ref_search_for_triangulation_pinhole_cams(c["k0"], c["d0"], c["s0"], c["ur0"], c["fva"], c["k1"], c["d1"], c["s1"], c["ur1"], c["fvb"],
                                                                     c["sf"], c["sg"], c["K1"], c["K2"], c["pose1"], c["pose2"], c["ori"], c["only_stereo"], c["coarse"])
        return live["r"]
    # key frame 1 at the origin: Cw = 0, C2 = t2, epipole = Pinhole::project(t2) (float operations in the order of Pinhole.cpp:43-49)
    t2 = c["pose2"][1]
    K2 = np.asarray(c["K2"], np.float32)
    ep = (K2[0] * t2[0] / t2[2] + K2[2], K2[1] * t2[1] / t2[2] + K2[3])
    if c["coarse"]:
        F = np.eye(3, dtype=np.float32)   # bCoarse: epipolarConstrain is never evaluated, no F12 exists on the reference side
    else:
        F = _P.value(f"m7cams/{name}/F12", lambda: run()[2]).reshape(3, 3)
        assert np.abs(F).max() > 0
    s0 = c["s0"].copy()
    s1 = c["s1"].copy()
    if c["only_stereo"]:   # :971-983 / :1002-1012: monocular keypoints are skipped on both sides
        s0 |= (c["ur0"] < 0).astype(np.uint8)
        s1 |= (c["ur1"] < 0).astype(np.uint8)
    on, om = ob.search_for_triangulation_pinhole(c["k0"], c["d0"], s0, c["ur0"], c["fva"], c["k1"], c["d1"], s1, c["ur1"], c["fvb"], c["sf"], c["sg"], F, ep,
                                                 c["coarse"], c["ori"], fma=True)
    _pin(f"m7cams/{name}", (on, om), lambda: run()[:2])
    if rb.matcher_available() and not c["coarse"] and f"m7cams/{name}/F12" in _P.gold:   # both builds (reference, adapter) must have used the same F12
        assert np.array_equal(_P.gold[f"m7cams/{name}/F12"].view(np.float32).reshape(3, 3), run()[2]), "F12 of this run differs from the compiled reference's"
    assert on > (5 if name == "forward_epipole_in_image" else 40), on


@pytest.mark.parametrize("case", [("tumvi_rig", 40, True, False), ("tumvi_rig_big_nodes", 3, False, False), ("tumvi_rig_coarse", 40, True, True)], ids=lambda c: c[0])
def test_search_for_triangulation_between_kannala_brandt_rigs(case):
    """M7 between two key frames of a KannalaBrandt8 stereo rig (TUM-VI's), the way LocalMapping::CreateNewMapPoints calls it: key frames with mvKeys | mvKeysRight,
    two KannalaBrandt8 camera objects each, Tcw and Trl -- nothing else.  Reference side: ORBmatcher.cc:907-1146 whole -- the four relative poses of :934-944 on
    the stand-in Sophus, camera objects and pose picked per pair (:1036-1069), the camera's epipolarConstrain (the shell forwards to the oracle's
    KannalaBrandt8::epipolarConstrain, itself pinned to the reference's text in test_oracle_geometry.py).  The oracle's form of the search gets the four poses that
    run used and must return the same pairs.  With ORBX_MATCHER_BACKEND=adapter the same call goes through the drop-in adapter, which must take its KannalaBrandt8
    route (the shell's host epipolarConstrain aborts in that build): poses from the adapter's Sophus, gate inside k_tri_kb8."""
    from orb_slam3_amd import synth
    name, nodes, ori, coarse = case
    rng = np.random.default_rng(sum(map(ord, name)))
    k1, nl1, d1, id1, k2, nl2, d2, id2, _, _, cams, poses = synth.make_fisheye_keyframes(rng, 300, with_poses=True)
    fv1, fv2 = FeatureVector.from_node_of_feature(id1 % nodes), FeatureVector.from_node_of_feature(id2 % nodes)
    s1 = (rng.random(len(k1)) < 0.2).astype(np.uint8)
    s2 = (rng.random(len(k2)) < 0.2).astype(np.uint8)
    sg = (np.array([1.2 ** i for i in range(8)], np.float32) ** 2).astype(np.float32)
    live = {}

    def run():
        if "r" not in live:
            live["r"] = rb.ref_search_for_triangulation_kb8_cams(k1, nl1, d1, s1, fv1, k2, nl2, d2, s2, fv2, sg, cams[0], cams[1], poses["pose1"], poses["pose2"],
                                                                 poses["trl"], ori, coarse)
        return live["r"]
    R12 = _P.value(f"m7kb8/{name}/R12", lambda: run()[2]).reshape(4, 3, 3)
    t12 = _P.value(f"m7kb8/{name}/t12", lambda: run()[3]).reshape(4, 3)
    on, om = ob.search_for_triangulation_kb8(k1, nl1, d1, s1, fv1, k2, nl2, d2, s2, fv2, sg, sg, cams, cams, R12, t12, coarse, ori)
    _pin(f"m7kb8/{name}", (on, om), lambda: run()[:2])
    if rb.matcher_available() and f"m7kb8/{name}/R12" in _P.gold:   # both builds (reference, adapter) must have formed the same relative poses
        assert np.array_equal(_P.gold[f"m7kb8/{name}/R12"].view(np.float32), run()[2].ravel()) and np.array_equal(_P.gold[f"m7kb8/{name}/t12"].view(np.float32), run()[3].ravel())
    hit = om >= 0
    assert on > 60 and (om[hit] >= nl2).sum() > 5 and (np.nonzero(hit)[0] >= nl1).sum() > 5, on   # matches in all camera pairings
    if not coarse:
        assert (id1[hit] == id2[om[hit]]).mean() > 0.9


def _fuse_queries(rng, k0, d0, sf, th):
    n = len(k0)
    u = (k0["x"] - 2.0 + rng.normal(0, 1.5, n)).astype(np.float32)
    v = (k0["y"] - 1.0 + rng.normal(0, 1.5, n)).astype(np.float32)
    inside = (u >= 0) & (u < W) & (v >= 0) & (v < H)
    u, v, lvl = u[inside], v[inside], k0["octave"][inside]
    z = _depths(rng, len(u))
    return dict(u=u, v=v, z=z, ur=(u - np.float32(1.0) / z).astype(np.float32), r=(np.float32(th) * sf[lvl]).astype(np.float32),
                level=lvl, desc=_noisy_copy(rng, d0[inside], 0.03))


def test_fuse(frames):
    """Fuse, ORBmatcher.cc:1148-1338 (chi2 gates 5.99 mono / 7.8 stereo on e2 * invSigma2, TH_LOW) and :1340-1455 (Sim3 form)."""
    k0, d0, k1, d1, tab = frames[1000]
    sf, isg = tab["scale"], tab["inv_sigma2"]
    rng = np.random.default_rng(51)
    ur = np.where(rng.random(len(k1)) < 0.5, k1["x"] - rng.uniform(0, 3, len(k1)), -1.0).astype(np.float32)
    for th, u_right in ((3.0, None), (4.0, ur), (2.5, ur)):     # LocalMapping.cc:738/760, LoopClosing.cc:2374
        q = _fuse_queries(rng, k0, d0, sf, th)
        grid, KF = _views(k1, d1, sf, u_right)
        obi, obd = ob.fuse_search(grid, d1, u_right, isg, q, fma=True)
        want = np.where(obd <= 50, obi, -1)
        _pin(f"fuse0/{th}/{u_right is not None}", ((want >= 0).sum(), want), lambda: rb.ref_fuse(KF, isg, q, th, 0))
        assert (want >= 0).sum() > 100
        obi, obd = ob.fuse_search(grid, d1, None, None, q, fma=True)
        want = np.where(obd <= 50, obi, -1)
        _pin(f"fuse1/{th}/{u_right is not None}", ((want >= 0).sum(), want), lambda: rb.ref_fuse(KF, isg, q, th, 1))
        assert (want >= 0).sum() > 100


def test_fuse_right_camera_fisheye(frames):
    """Fuse(pKF, vpMapPoints, th, bRight=True), ORBmatcher.cc:1148-1337 on a fisheye-stereo key frame: right camera's pose and grid,
    mvKeysRight octaves, monocular chi2 gate (mvuRight is -1 throughout, Frame.cc:1137), fused index reported as idx + NLeft."""
    k0, d0, k1, d1, tab = frames[1000]
    sf, isg = tab["scale"], tab["inv_sigma2"]
    rng = np.random.default_rng(71)
    kl, dl, kr, dr = k1, d1, k0, d0            # left = frame 1, right = frame 0
    desc = np.concatenate([dl, dr])
    bounds = np.array([0.0, W, 0.0, H], np.float32)
    gr = ob.OracleGrid(kr, 0.0, float(W), 0.0, float(H))
    for th in (3.0, 4.0):
        q = _fuse_queries(rng, kr, dr, sf, th)      # map points that project near the RIGHT camera's features
        q["u"] = (q["u"] + 2.0).astype(np.float32); q["v"] = (q["v"] + 1.0).astype(np.float32)
        q["ur"] = (q["u"] - np.float32(1.0) / q["z"]).astype(np.float32)
        obi, obd = ob.fuse_search(gr, dr, None, isg, q, fma=True)
        want = np.where(obd <= 50, obi + len(kl), -1)
        _pin(f"fuse_right/{th}", ((want >= 0).sum(), want), lambda: rb.ref_fuse_right(kl, kr, desc, bounds, sf, isg, q, th))
        assert (want >= 0).sum() > 100


def test_search_for_triangulation_fisheye_pairings(frames):
    """M7 between two fisheye-stereo key frames, ORBmatcher.cc:907-1146 with the four camera pairings of :1036-1069: no epipole test
    (pKF1->mpCamera2), no stereo flags, per pair the camera objects and T12 of (left|right, left|right) -- the stand-in cameras
    refuse a pair evaluated with the wrong objects or the wrong translation."""
    k0, d0, k1, d1, _ = frames[1000]
    rng = np.random.default_rng(73)
    n1, n2 = len(k0), len(k1)
    p0, p1 = rng.permutation(n1), rng.permutation(n2)   # shuffled, so that corresponding features fall into all four camera pairings
    k0, d0, k1, d1 = k0[p0], d0[p0], k1[p1], d1[p1]
    nl1, nl2 = int(0.6 * n1), int(0.55 * n2)       # features beyond n_left belong to the right cameras
    na, nb = _bow_nodes(rng, k0, k1, 60)
    fva, fvb = FeatureVector.from_node_of_feature(na), FeatureVector.from_node_of_feature(nb)
    s0 = (rng.random(n1) < 0.3).astype(np.uint8)
    s1 = (rng.random(n2) < 0.3).astype(np.uint8)
    table = (rng.random((n1, n2)) < 0.7).astype(np.uint8)
    for ori, coarse in ((True, False), (False, False), (True, True)):
        pred = None if coarse else (lambda i, j: table[i, j])
        on, om = ob.search_for_triangulation(d0, k0["angle"], s0, fva, d1, k1["angle"], s1, fvb, ori, pred)
        _pin(f"m7fisheye/{ori}/{coarse}", (on, om),
             lambda: rb.ref_search_for_triangulation_fisheye(d0, k0["angle"], s0, nl1, fva, d1, k1["angle"], s1, nl2, fvb, ori, table, coarse))
        assert on > 50 and (om[nl1:] >= nl2).sum() > 5 and (om[:nl1] >= nl2).sum() > 5 and ((om[nl1:] >= 0) & (om[nl1:] < nl2)).sum() > 5


def test_search_by_sim3(frames):
    """SearchBySim3, ORBmatcher.cc:1457-1674 = two gate-less fuse searches (best <= TH_HIGH, octave gate [l-1,l]) + mutual
    agreement, composed here exactly as orb_slam3_amd composes it over orbx_fuse_search."""
    k0, d0, k1, d1, tab = frames[1000]
    sf = tab["scale"]
    rng = np.random.default_rng(61)
    th = 7.5
    g0, K0 = _views(k0, d0, sf)
    g1, K1 = _views(k1, d1, sf)

    def side(k, d, dx, dy):
        n = len(k)
        x = (k["x"] + dx + rng.normal(0, 1.0, n)).astype(np.float32)
        y = (k["y"] + dy).astype(np.float32)
        valid = rng.choice([0, 1, 1, 1, 2], n).astype(np.uint8)
        valid[(x < 0) | (x >= W) | (y < 0) | (y >= H)] = 0
        return dict(valid=valid, x=x, y=y, level=k["octave"], desc=_noisy_copy(rng, d, 0.03))
    s0, s1 = side(k0, d0, -2.0, -1.0), side(k1, d1, 2.0, 1.0)

    def one_way(s, grid, dtab):
        q = dict(u=s["x"], v=s["y"], ur=np.zeros(len(s["x"]), np.float32), r=(np.float32(th) * sf[s["level"]]).astype(np.float32),
                 level=s["level"], desc=s["desc"])
        bi, bd = ob.fuse_search(grid, dtab, None, None, q, fma=True)
        return np.where((bd <= 100) & (s["valid"] == 1), bi, -1)
    m1, m2 = one_way(s0, g1, d1), one_way(s1, g0, d0)
    want = np.array([m1[i] if (m1[i] >= 0 and m2[m1[i]] == i) else -1 for i in range(len(k0))], np.int32)
    _pin("sim3", ((want >= 0).sum(), want), lambda: rb.ref_search_by_sim3(K0, K1, s0, s1, th))
    assert (want >= 0).sum() > 100

    # the product's host-side composition (orb_slam3_amd.ORBmatcher.SearchBySim3) with its device search swapped for the oracle's:
    # the agreement / validity logic above the C ABI must give the same answer (the device search itself is a -m gpu test)
    from orb_slam3_amd.matcher import ORBmatcher, FrameView
    host = object.__new__(ORBmatcher)
    grids = {id(k0): g0, id(k1): g1}
    host.FuseSearch = lambda KF, q, isg=None, strict_fp=False: ob.fuse_search(grids[id(KF.keypoints_un)], KF.descriptors, None, None, q)
    fv0, fv1 = FrameView(k0, d0, 0.0, float(W), 0.0, float(H), sf), FrameView(k1, d1, 0.0, float(W), 0.0, float(H), sf)
    p0, p1 = dict(s0, u=s0["x"], v=s0["y"]), dict(s1, u=s1["x"], v=s1["y"])
    n, m12 = ORBmatcher.SearchBySim3(host, fv0, fv1, p0, p1, th)
    assert n == (want >= 0).sum() and np.array_equal(m12, want)
