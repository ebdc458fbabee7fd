"""Oracle restatements of the candidate-generation pre-passes.

* Frame::isInFrustum (+ MapPoint::PredictScale, Pinhole::project): the oracle against the reference's own text compiled where it
  lies (oracle/_ref/libfrustum_ref.so: stand-in float Eigen, operations rounded as written), bit for bit, live or against the
  committed outputs of that library (tests/golden/frustum_ref.npz).
* cv::undistortPoints [OCV-recalled]: parity UNPINNED (no OpenCV here); checked against the defining property instead -- pushing the
  undistorted point through the forward radial-tangential model the reference itself spells out (Frame::ProjectPointDistort,
  Frame.cc:577-645) must give back the distorted pixel."""
import numpy as np
import pytest

from oracle import oracle_binding as ob
from oracle import ref_binding as rb
from _pin import Pinner

_P = Pinner("frustum_ref.npz", rb.frustum_available())


@pytest.fixture(scope="module", autouse=True)
def _write_golden():
    yield
    _P.finish()


def frustum_case(seed, n=4000):
    rng = np.random.default_rng(seed)
    a, b, c = rng.uniform(-0.3, 0.3, 3)       # a pose with a real rotation
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    Rcw = (Rz @ Ry @ Rx).astype(np.float32)
    tcw = rng.uniform(-1, 1, 3).astype(np.float32)
    Ow = (-(Rcw.astype(np.float64).T @ tcw.astype(np.float64))).astype(np.float32)
    cam = (458.654, 457.296, 367.215, 248.375, 47.9)             # EuRoC cam0 intrinsics, bf
    bounds = np.array([-10.5, 760.25, -8.0, 488.5], np.float32)  # undistorted image bounds stick out of the image
    pos = rng.uniform(-6, 6, (n, 3)).astype(np.float32)
    pos[:, 2] = rng.uniform(-2, 12, n)                            # some behind the camera
    normal = rng.normal(0, 1, (n, 3))
    normal = (normal / np.linalg.norm(normal, axis=1, keepdims=True)).astype(np.float32)
    towards = (Ow[None, :] - pos)
    towards /= np.linalg.norm(towards, axis=1, keepdims=True)
    flip = rng.random(n) < 0.6                                    # most normals roughly face the camera
    normal[flip] = -(towards[flip] + rng.normal(0, 0.3, (flip.sum(), 3))).astype(np.float32)
    normal = (normal / np.linalg.norm(normal, axis=1, keepdims=True)).astype(np.float32)
    dist = np.linalg.norm(pos - Ow, axis=1)
    max_d = (dist * rng.uniform(0.7, 4.0, n)).astype(np.float32)
    min_d = (max_d / rng.uniform(1.5, 4.3, n)).astype(np.float32)
    return Rcw, tcw, Ow, cam, bounds, np.float32(np.log(1.2)), 8, 0.5, pos, normal, min_d, max_d


KEYS = ("in_view", "proj_x", "proj_y", "proj_xr", "depth", "level", "view_cos")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_is_in_frustum_equals_reference_text(seed):
    args = frustum_case(seed)
    o = ob.is_in_frustum(*args)
    # fields the reference leaves untouched on rejection are whatever the MapPoint held before: compare them only where in_view
    def flat(d):
        iv = d["in_view"].astype(bool)
        return [d["in_view"], d["proj_x"], d["proj_y"]] + [np.where(iv, d[k], 0) for k in KEYS[3:]]
    _P.pin(f"frustum/{seed}", flat(o), lambda: flat(rb.ref_is_in_frustum(*args)))
    iv = o["in_view"].astype(bool)
    assert 200 < iv.sum() < len(iv) - 200
    assert len(np.unique(o["level"][iv])) >= 6 and (o["proj_x"][~iv] == -1).sum() > 100 and ((o["proj_x"] != -1) & ~iv).sum() > 100


TUMVI_L = (190.978477, 190.973307, 254.931706, 256.897442, 0.0034823894, 0.0007150348, -0.0020532361, 0.0002029367)   # Examples/Stereo/TUM-VI.yaml cam1
TUMVI_R = (190.442369, 190.434438, 252.598164, 254.917230, 0.0034003171, 0.0017669271, -0.0026631290, 0.0003299517)   # cam2
KEYS_F = ("in_view", "proj_x", "proj_y", "depth", "level", "view_cos")


def fisheye_case(seed, n=4000):
    """A fisheye rig frame: pose, mTrl (right-from-left, ~10 cm baseline with a small rotation), map points spread over more than a hemisphere."""
    Rcw, tcw, Ow, _, _, lsf, nl, cosl, pos, normal, mn, mx = frustum_case(seed, n)
    rng = np.random.default_rng(1000 + seed)
    pos = pos.copy()
    pos[:, :2] *= rng.uniform(0.5, 3.0, (n, 1)).astype(np.float32)   # wide field of view
    a = rng.uniform(-0.02, 0.02, 3)
    Rrl = (np.array([[1, -a[2], a[1]], [a[2], 1, -a[0]], [-a[1], a[0], 1]]) @ np.eye(3)).astype(np.float32)
    trl = np.array([-0.101, 0.002, 0.001], np.float32) + rng.normal(0, 1e-3, 3).astype(np.float32)
    tlr = (-(Rrl.astype(np.float64).T @ trl.astype(np.float64))).astype(np.float32)
    Rwc = np.ascontiguousarray(Rcw.T)
    bounds = np.array([0.0, 512.0, 0.0, 512.0], np.float32)
    return dict(Rcw=Rcw, tcw=tcw, Ow=Ow, Rwc=Rwc, Rrl=Rrl, trl=trl, tlr=tlr, bounds=bounds, lsf=lsf, nl=nl, cosl=cosl, pos=pos, normal=normal, mn=mn, mx=mx)


def fisheye_views(c, pin_name=None):
    """(R, t, twc, params) of the left and right camera: what Frame.cc:1172-1186 compute, taken from the reference text (live or committed)."""
    views = []
    for right, prm in ((0, TUMVI_L), (1, TUMVI_R)):
        v = _P.value(f"{pin_name}/view{right}", lambda: rb.ref_is_in_frustum_checks(c["Rcw"], c["tcw"], c["Ow"], c["Rwc"], c["Rrl"], c["trl"], c["tlr"], TUMVI_L, TUMVI_R,
                                                                                   right, c["bounds"], c["lsf"], c["nl"], c["cosl"], c["pos"][:1], c["normal"][:1],
                                                                                   c["mn"][:1], c["mx"][:1])[1])
        views.append((v[:9], v[9:12], v[12:15], np.array(prm, np.float32)))
    return views


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_is_in_frustum_checks_equals_reference_text(seed):
    """Frame::isInFrustumChecks + KannalaBrandt8::project (fisheye rigs: TUM-VI): the oracle against the reference's text, both cameras, bit for bit."""
    c = fisheye_case(seed)
    views = fisheye_views(c, f"fisheye/{seed}")
    for right in (0, 1):
        o = ob.is_in_frustum_checks(views[right], c["bounds"], c["lsf"], c["nl"], c["cosl"], c["pos"], c["normal"], c["mn"], c["mx"])
        def flat(d):
            iv = d["in_view"].astype(bool)   # fields the reference leaves untouched on rejection: compared only where in_view
            return [d["in_view"]] + [np.where(iv, d[k], 0) for k in KEYS_F[1:]]
        _P.pin(f"fisheye/{seed}/{right}", flat(o), lambda: flat(rb.ref_is_in_frustum_checks(c["Rcw"], c["tcw"], c["Ow"], c["Rwc"], c["Rrl"], c["trl"], c["tlr"], TUMVI_L,
                                                                                           TUMVI_R, right, c["bounds"], c["lsf"], c["nl"], c["cosl"], c["pos"],
                                                                                           c["normal"], c["mn"], c["mx"])[0]))
        iv = o["in_view"].astype(bool)
        assert 200 < iv.sum() < len(iv) - 200 and len(np.unique(o["level"][iv])) >= 6
        assert np.all(o["level"][~iv] == -1)


def kb8_pairs(seed, n=3000):
    """Keypoint pairs of two fisheye cameras looking at common 3-D points (+ pixel noise of several sizes, some pairs with hardly any parallax):
    every return path of KannalaBrandt8::TriangulateMatches is taken."""
    rng = np.random.default_rng(500 + seed)
    a = rng.uniform(-0.05, 0.05, 3)
    R12 = np.array([[1, -a[2], a[1]], [a[2], 1, -a[0]], [-a[1], a[0], 1]], np.float32)
    t12 = (np.array([0.1, 0.0, 0.0]) + rng.normal(0, 0.02, 3)).astype(np.float32)
    X1 = np.stack([rng.uniform(-2, 2, n), rng.uniform(-2, 2, n), rng.uniform(0.8, 7, n)], axis=1)
    X2 = (R12.astype(np.float64).T @ (X1 - t12.astype(np.float64)).T).T
    def proj(prm, X):
        th = np.arctan2(np.hypot(X[:, 0], X[:, 1]), X[:, 2]); psi = np.arctan2(X[:, 1], X[:, 0])
        r = th + prm[4] * th ** 3 + prm[5] * th ** 5 + prm[6] * th ** 7 + prm[7] * th ** 9
        return np.stack([prm[0] * r * np.cos(psi) + prm[2], prm[1] * r * np.sin(psi) + prm[3]], axis=1)
    noise = rng.choice([0.3, 1.5, 6.0], n)[:, None]
    xy1 = (proj(TUMVI_L, X1) + noise * rng.uniform(-1, 1, (n, 2))).astype(np.float32)
    xy2 = (proj(TUMVI_R, X2) + noise * rng.uniform(-1, 1, (n, 2))).astype(np.float32)
    same = rng.random(n) < 0.15
    xy2[same] = xy1[same]
    lev = 1.44 ** rng.integers(0, 8, (n, 2))
    return R12, t12, xy1, xy2, lev[:, 0].astype(np.float32), lev[:, 1].astype(np.float32)


@pytest.mark.parametrize("seed", [1, 2])
def test_kb8_epipolar_constrain_equals_reference_text(seed):
    """KannalaBrandt8::epipolarConstrain = TriangulateMatches > 0.0001 (KannalaBrandt8.cpp:216-221, 305-368; Triangulate :387-400; unproject :107-142): the oracle
    against the reference's text for all of it -- over the stand-in Eigen types, whose JacobiSVD IS the oracle's restatement of Eigen's algorithm (Eigen is absent:
    the decomposition itself stays unpinned; the SVD is checked on its defining properties below)."""
    R12, t12, xy1, xy2, s1, s2 = kb8_pairs(seed)
    ok, val = ob.kb8_epipolar_constrain(TUMVI_L, TUMVI_R, xy1, xy2, R12, t12, s1, s2)
    rays = np.concatenate([ob.kb8_unproject(TUMVI_L, xy1), ob.kb8_unproject(TUMVI_R, xy2)], axis=1)
    _P.pin(f"kb8_epipolar/{seed}", [ok, val, rays], lambda: list(rb.ref_kb8_triangulate_matches(TUMVI_L, TUMVI_R, xy1, xy2, R12, t12, s1, s2)))
    codes = {c: int((val == c).sum()) for c in (-1, -2, -4, -5)}
    assert ok.sum() > 500 and min(codes.values()) > 10, (int(ok.sum()), codes)


def test_restated_jacobi_svd_properties():
    """The restated Eigen::JacobiSVD<Matrix4f>: V orthonormal, singular values descending and equal to numpy's, A V = U S column norms -- on random and on
    rank-deficient matrices (the triangulation's A has a one-dimensional null space: its last column of V is what the reference reads)."""
    rng = np.random.default_rng(0)
    L = ob.lib()
    for t in range(500):
        A = rng.normal(0, 1, (4, 4)).astype(np.float32)
        if t % 3 == 0:
            A[3] = 0.5 * A[0] + A[1]
        V, sv = np.zeros(16, np.float32), np.zeros(4, np.float32)
        L.orbo_eigen_jacobi_svd4_V(ob._p(A), ob._p(V), ob._p(sv))
        V = V.reshape(4, 4).astype(np.float64)
        s = np.linalg.svd(A.astype(np.float64), compute_uv=False)
        assert np.all(np.diff(sv) <= 0) and np.abs(s - sv).max() < 2e-6 * s.max()
        assert np.abs(V.T @ V - np.eye(4)).max() < 2e-6
        assert np.abs(np.linalg.norm(A.astype(np.float64) @ V, axis=0) - sv).max() < 3e-6 * s.max()


def _distort(xy_un, cam, dist):
    """Frame::ProjectPointDistort's forward model (Frame.cc:612-636) in float64."""
    fx, fy, cx, cy = cam
    k1, k2, p1, p2, k3 = dist
    x, y = (xy_un[:, 0] - cx) / fx, (xy_un[:, 1] - cy) / fy
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
    xd = x * rad + (2 * p1 * x * y + p2 * (r2 + 2 * x * x))
    yd = y * rad + (p1 * (r2 + 2 * y * y) + 2 * p2 * x * y)
    return np.stack([xd * fx + cx, yd * fy + cy], axis=1)


@pytest.mark.parametrize("cam,dist,size", [((458.654, 457.296, 367.215, 248.375), (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0), (752, 480)),   # EuRoC cam0
                                            ((517.3, 516.5, 318.6, 255.3), (0.2624, -0.9531, -0.0054, 0.0026, 1.1633), (640, 480))])                    # TUM1
def test_undistort_points_inverts_the_forward_model(cam, dist, size):
    rng = np.random.default_rng(5)
    xy = np.stack([rng.uniform(0, size[0], 5000), rng.uniform(0, size[1], 5000)], axis=1).astype(np.float32)
    un = ob.undistort_points(xy, cam, dist)
    back = _distort(un.astype(np.float64), cam, dist)
    err = np.abs(back - xy).max(axis=1)
    # cv::undistortPoints stops after 5 fixed-point iterations: converged to thousandths of a pixel mid-image, tenths in the corners
    assert np.percentile(err, 99) < 0.5 and np.median(err) < 2e-2 and np.percentile(err, 25) < 2e-3
    assert np.abs(un - xy).max() > 3.0                               # and the distortion is not a no-op
    size = (752, 480)
    b = ob.image_bounds(752, 480, cam, dist)
    corners = ob.undistort_points(np.array([[0, 0], [752, 0], [0, 480], [752, 480]], np.float32), cam, dist)
    assert b[0] == min(corners[0, 0], corners[2, 0]) and b[1] == max(corners[1, 0], corners[3, 0])
    assert b[2] == min(corners[0, 1], corners[1, 1]) and b[3] == max(corners[2, 1], corners[3, 1])
    assert np.array_equal(ob.image_bounds(752, 480, cam, (0.0, 0, 0, 0, 0)), np.array([0, 752, 0, 480], np.float32))
