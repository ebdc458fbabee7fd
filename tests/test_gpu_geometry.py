"""GPU parity of the candidate-generation pre-passes against the oracle: Frame::isInFrustum (oracle pinned to the
reference text), Frame::UndistortKeyPoints / ComputeImageBounds (cv::undistortPoints restated, unpinned), and the fused batch path
extract -> undistort -> isInFrustum -> SearchByProjection(Frame, MapPoints), nothing leaving the device in between.

Tolerance: every compared float is required to be BIT-IDENTICAL to the oracle (the kernels round each operation as the reference
text writes it), except the predicted level, which goes through logf (libm vs the device's double log): it may differ where
log(ratio)/logScaleFactor lies within 1e-5 of an integer -- asserted, and in practice it never differs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EUROC = (458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0, 47.9)
KEYS = ("in_view", "proj_x", "proj_y", "proj_xr", "depth", "level", "view_cos")


def _same(got, want, args):
    iv = want["in_view"].astype(bool)
    assert np.array_equal(got["in_view"], want["in_view"])
    for k in ("proj_x", "proj_y"):
        assert got[k].tobytes() == want[k].tobytes(), k
    for k in ("proj_xr", "depth", "view_cos"):
        assert got[k][iv].tobytes() == want[k][iv].tobytes(), k
    diff = np.nonzero(got["level"][iv] != want["level"][iv])[0]
    if len(diff):   # only at an integer boundary of log(ratio) / logScaleFactor
        pos, Ow, mx = args[8][iv][diff], args[2], args[11][iv][diff]
        q = np.log(mx.astype(np.float64) / np.linalg.norm(pos - Ow, axis=1)) / float(args[5])
        assert np.all(np.abs(q - np.rint(q)) < 1e-5)
    return len(diff)


def test_is_in_frustum_equals_oracle(oracle):
    import orb_slam3_amd as osa
    from test_oracle_geometry import frustum_case
    m = osa.ORBmatcher(0.8, True)
    for seed in (1, 2, 3, 4):
        args = frustum_case(seed, n=20000)
        Rcw, tcw, Ow, cam, bounds, lsf, nl, cosl, pos, normal, mn, mx = args
        want = oracle.is_in_frustum(*args)
        got = m.isInFrustum(cam[:4] + (0, 0, 0, 0, 0, cam[4]), (Rcw, tcw, Ow), bounds, lsf, nl, cosl, pos, normal, mn, mx)
        _same(got, want, args)
        assert 1000 < want["in_view"].sum() < 19000


def test_is_in_frustum_checks_equals_oracle(oracle):
    """Frame::isInFrustumChecks + KannalaBrandt8::project for both cameras of a fisheye rig (Frame.cc:1168, KannalaBrandt8.cpp:67) on the device.  atan2f is
    glibc's bit for bit; cos / sin of the azimuth go through the device's double functions: proj_x / proj_y are allowed ONE float ulp (counted: in practice 0),
    a decision may only flip where the projection sits within that ulp of an image bound."""
    import orb_slam3_amd as osa
    from test_oracle_geometry import fisheye_case, fisheye_views
    m = osa.ORBmatcher(0.8, True)
    n_diff = 0
    for seed in (1, 2, 3):
        c = fisheye_case(seed, n=20000 if seed == 1 else 4000)   # (the views of seeds 1-3 are committed goldens of the reference text; n only sets the map points)
        views = fisheye_views(fisheye_case(seed), f"fisheye/{seed}")
        got = m.isInFrustumChecks(views, c["bounds"], c["lsf"], c["nl"], c["cosl"], c["pos"], c["normal"], c["mn"], c["mx"])
        for right in (0, 1):
            want = oracle.is_in_frustum_checks(views[right], c["bounds"], c["lsf"], c["nl"], c["cosl"], c["pos"], c["normal"], c["mn"], c["mx"])
            g = {k: v[right] for k, v in got.items()}
            flip = np.nonzero(g["in_view"] != want["in_view"])[0]
            assert len(flip) <= 2, len(flip)
            both = (g["in_view"] & want["in_view"]).astype(bool)
            for k in ("proj_x", "proj_y"):
                ulp = np.abs(g[k][both].view(np.int32).astype(np.int64) - want[k][both].view(np.int32).astype(np.int64))
                assert ulp.max(initial=0) <= 1, (k, ulp.max())
                n_diff += int((ulp != 0).sum())
            for k in ("depth", "view_cos"):
                assert g[k][both].tobytes() == want[k][both].tobytes(), k
            lv = np.nonzero(g["level"][both] != want["level"][both])[0]
            assert len(lv) <= 2
            assert np.all(g["level"][~g["in_view"].astype(bool)] == -1)
            assert 200 < want["in_view"].sum() < len(both) - 200
    assert n_diff <= 4, n_diff


def test_undistort_and_bounds_equal_oracle(oracle):
    import orb_slam3_amd as osa
    m = osa.ORBmatcher(0.8, True)
    rng = np.random.default_rng(2)
    k = np.zeros(5000, oracle.KP_DTYPE)
    k["x"], k["y"] = rng.uniform(0, 752, 5000), rng.uniform(0, 480, 5000)
    k["octave"], k["angle"], k["size"], k["response"], k["class_id"] = rng.integers(0, 8, 5000), rng.uniform(0, 360, 5000), 31, 40, -1
    for cam in (EUROC, (517.3, 516.5, 318.6, 255.3, 0.2624, -0.9531, -0.0054, 0.0026, 1.1633, 40.0)):
        un = m.UndistortKeyPoints(cam, k)
        want = oracle.undistort_points(np.stack([k["x"], k["y"]], axis=1), cam[:4], cam[4:9])
        assert un["x"].tobytes() == want[:, 0].tobytes() and un["y"].tobytes() == want[:, 1].tobytes()
        for f in ("octave", "angle", "size", "response", "class_id"):
            assert np.array_equal(un[f], k[f])
        assert np.array_equal(m.ComputeImageBounds(cam, 752, 480), oracle.image_bounds(752, 480, cam[:4], cam[4:9]))
    flat = EUROC[:4] + (0.0, 0.0, 0.0, 0.0, 0.0, 47.9)          # k1 == 0: mvKeysUn = mvKeys (Frame.cc:749-753)
    assert m.UndistortKeyPoints(flat, k).tobytes() == k.tobytes()


def test_fused_batch_extract_undistort_frustum_search(oracle, canvas1):
    """Tracking::SearchLocalPoints on a resident batch with a distorting camera: extraction, UndistortKeyPoints, isInFrustum for every
    (frame pose, map point) and SearchByProjection(Frame, MapPoints) all on the device; compared with the oracle chain frame by frame."""
    import torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    B, W, H, NF, n_mp = 5, 752, 480, 1000, 6000
    frames = np.stack([synth.frame_from_canvas(canvas1, t, W, H, 1000 + t) for t in range(B)])
    ex = osa.ORBextractor(NF, 1.2, 8, 20, 7)
    ex.set_camera(EUROC)
    d_frames = torch.from_numpy(frames).cuda()
    ex.extract_batch_device(d_frames.data_ptr(), B, W, H, W, W * H, (0, 1000))
    feats = [ex.download(f) for f in range(B)]
    kun = [ex.download_keypoints_un(f) for f in range(B)]
    cam4, dist5 = EUROC[:4], EUROC[4:9]
    bounds = oracle.image_bounds(W, H, cam4, dist5)
    for f in range(B):   # the device's mvKeysUn are the oracle's
        want = oracle.undistort_points(np.stack([feats[f][1]["x"], feats[f][1]["y"]], axis=1), cam4, dist5)
        assert kun[f]["x"].tobytes() == want[:, 0].tobytes() and kun[f]["y"].tobytes() == want[:, 1].tobytes()
    # a local map in front of the cameras: points that back-project from features of frame 0 at random depths
    rng = np.random.default_rng(11)
    src = rng.integers(0, len(kun[0]), n_mp)
    z = rng.uniform(2.0, 9.0, n_mp)
    pos = np.stack([(kun[0]["x"][src] - cam4[2]) / cam4[0] * z, (kun[0]["y"][src] - cam4[3]) / cam4[1] * z, z], axis=1).astype(np.float32)
    pos += rng.normal(0, 0.01, pos.shape).astype(np.float32)
    normal = pos / np.linalg.norm(pos, axis=1, keepdims=True)     # mean viewing direction: from the camera towards the point
    normal = (normal + rng.normal(0, 0.2, normal.shape)).astype(np.float32)
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    dist0 = np.linalg.norm(pos, axis=1)
    max_d = (dist0 * 1.2 ** kun[0]["octave"][src] * rng.uniform(0.9, 1.1, n_mp)).astype(np.float32)
    min_d = (max_d / 1.2 ** 7).astype(np.float32)
    desc = feats[0][2][src].copy()
    desc ^= np.packbits(rng.random((n_mp, 256)) < 0.03, axis=1, bitorder="little")
    poses = []
    for f in range(B):   # the camera drifts a little from frame to frame
        a = 0.004 * f
        Rcw = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
        tcw = np.array([0.01 * f, -0.005 * f, 0.02 * f], np.float32)
        Ow = (-(Rcw.astype(np.float64).T @ tcw.astype(np.float64))).astype(np.float32)
        poses.append((Rcw, tcw, Ow))
    dev = dict(pos=torch.from_numpy(pos).cuda(), normal=torch.from_numpy(np.ascontiguousarray(normal, np.float32)).cuda(),
               mn=torch.from_numpy(min_d).cuda(), mx=torch.from_numpy(max_d).cuda(), desc=torch.from_numpy(desc).cuda())
    o = dict(in_view=torch.zeros((B, n_mp), dtype=torch.uint8, device="cuda"), proj_x=torch.zeros((B, n_mp), device="cuda"),
             proj_y=torch.zeros((B, n_mp), device="cuda"), proj_xr=torch.zeros((B, n_mp), device="cuda"), depth=torch.zeros((B, n_mp), device="cuda"),
             level=torch.zeros((B, n_mp), dtype=torch.int32, device="cuda"), view_cos=torch.zeros((B, n_mp), device="cuda"))
    cap = ex.batch_view().cap
    d_match = torch.full((B, cap), -7, dtype=torch.int32, device="cuda")
    d_nm = torch.full((B,), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ex.frustum_batch_device(EUROC, poses, 0.5, n_mp, dev["pos"].data_ptr(), dev["normal"].data_ptr(), dev["mn"].data_ptr(), dev["mx"].data_ptr(),
                            *[o[k].data_ptr() for k in KEYS])
    ex.search_mappoints_batch_device(n_mp, o["proj_x"].data_ptr(), o["proj_y"].data_ptr(), o["level"].data_ptr(), o["view_cos"].data_ptr(),
                                     o["in_view"].data_ptr(), dev["desc"].data_ptr(), desc_frame_stride=0, th=3.0, nnratio=0.8,
                                     d_match=d_match.data_ptr(), d_nmatches=d_nm.data_ptr())
    ex.sync()
    match, nm = d_match.cpu().numpy(), d_nm.cpu().numpy()
    sf = ex.GetScaleFactors()
    lsf = np.float32(np.log(np.float32(1.2)))
    for f in range(B):
        args = (poses[f][0], poses[f][1], poses[f][2], cam4 + (EUROC[9],), bounds, lsf, 8, 0.5, pos, normal, min_d, max_d)
        want = oracle.is_in_frustum(*args)
        got = {k: o[k][f].cpu().numpy() for k in KEYS}
        assert _same(got, want, args) == 0
        assert want["in_view"].sum() > 2000
        mp = dict(proj_x=want["proj_x"], proj_y=want["proj_y"], proj_xr=want["proj_xr"], level=want["level"], view_cos=want["view_cos"], desc=desc,
                  in_view=want["in_view"], has_obs=np.ones(n_mp, np.uint8))
        grid = oracle.OracleGrid(kun[f], float(bounds[0]), float(bounds[1]), float(bounds[2]), float(bounds[3]))
        on, ofm = oracle.search_by_projection_mappoints(grid, feats[f][2], sf, mp, 3.0, 0.8)
        assert nm[f] == on and np.array_equal(match[f, :len(kun[f])], ofm), f
        assert on > 150
