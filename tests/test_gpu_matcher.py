"""GPU parity tests of the matchers (through the C ABI) vs the CPU oracle: bit-exact index pairs / distances."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand_desc(rng, n):
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


def _noisy_copy(rng, d, p):
    flip = (rng.random((len(d), 256)) < p)
    return d ^ np.packbits(flip, axis=1, bitorder="little")


def test_hamming_csr_and_best2(oracle):
    import orb_slam3_amd as osa
    rng = np.random.default_rng(0)
    m = osa.ORBmatcher()
    T = _rand_desc(rng, 1500)
    Q = _noisy_copy(rng, T[rng.integers(0, 1500, 400)], 0.1)
    Q[5] = T[7]
    lens = rng.integers(0, 90, 400)
    lens[3] = 0
    lens[10] = 200
    row_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cand = rng.integers(0, 1500, row_ptr[-1]).astype(np.int32)
    # inject exact ties: duplicate candidates
    cand[row_ptr[10] + 5] = cand[row_ptr[10] + 50]
    dist = m.hamming_csr(Q, T, row_ptr, cand)
    want = np.array([oracle.descriptor_distance(Q[i], T[cand[k]]) for i in range(400) for k in range(row_ptr[i], row_ptr[i + 1])])
    assert np.array_equal(dist, want)
    bp, bd, sp, sd = m.hamming_best2_csr(Q, T, row_ptr, cand)
    for i in range(400):
        ds = want[row_ptr[i]:row_ptr[i + 1]]
        b, b2, p, p2 = 256, 256, -1, -1
        for k, d in enumerate(ds):  # the reference's sequential scan (ORBmatcher.cc:103-119)
            if d < b:
                b2, p2, b, p = b, p, d, k
            elif d < b2:
                b2, p2 = d, k
        assert (bp[i], bd[i], sp[i], sd[i]) == (p, b, p2, b2), i


def test_descriptor_distance_known_answers(oracle):
    import orb_slam3_amd as osa
    m = osa.ORBmatcher()
    z = np.zeros((1, 32), np.uint8)
    f = np.full((1, 32), 255, np.uint8)
    one = z.copy()
    one[0, 17] = 0x10
    rp = np.array([0, 3], np.int32)
    T = np.concatenate([z, f, one])
    d = m.hamming_csr(z, T, rp, np.array([0, 1, 2], np.int32))
    assert list(d) == [0, 256, 1]


def test_knn2_with_ties(oracle):
    import orb_slam3_amd as osa
    rng = np.random.default_rng(1)
    m = osa.ORBmatcher()
    T = _rand_desc(rng, 2300)   # > one LDS tile
    T[100] = T[2200]            # exact duplicates -> distance ties, lower train index must win
    Q = _noisy_copy(rng, T[rng.integers(0, 2300, 700)], 0.05)
    Q[0] = T[100]
    idx, dist = m.knn2(Q, T)
    oi, od = oracle.knn2(Q, T)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    idx, dist = m.knn2(Q[:3], T[:1])   # fewer than k train rows
    oi, od = oracle.knn2(Q[:3], T[:1])
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)


def _extract_pair(w=752, h=480, nf=1000):
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    canvas = synth.make_canvas(3)
    left, right = synth.make_stereo_pair(3, 0, w, h, canvas)
    exl = osa.ORBextractor(nf, 1.2, 8, 20, 7)
    exr = osa.ORBextractor(nf, 1.2, 8, 20, 7)
    _, kl, dl = exl(left, None, (0, 0))
    _, kr, dr = exr(right, None, (0, 0))
    return exl, exr, kl, dl, kr, dr


def test_stereo_rowband_and_full_stereo(oracle):
    """BASELINE config 3 (KITTI shape 1241x376, nFeatures=2000): Frame::ComputeStereoMatches."""
    import orb_slam3_amd as osa
    exl, exr, kl, dl, kr, dr = _extract_pair(1241, 376, 2000)
    m = osa.ORBmatcher()
    sf, isf = exl.GetScaleFactors(), exl.GetInverseScaleFactors()
    pyl = [np.ascontiguousarray(p) for p in exl.mvImagePyramid]
    pyr = [np.ascontiguousarray(p) for p in exr.mvImagePyramid]
    bf, b = 0.53716 * 718.856, 0.53716
    on, our, odepth, obi, obd = oracle.compute_stereo_matches(kl, dl, kr, dr, sf, isf, pyl, pyr, bf, b)
    bi, bd = m.stereo_rowband(kl, dl, kr, dr, sf, 376, 0.0, bf / b)
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd)
    assert (bi >= 0).sum() > 200
    n, ur, depth = m.ComputeStereoMatches(kl, dl, kr, dr, sf, isf, pyl, pyr, bf, b)
    assert n == on and ur.tobytes() == our.tobytes() and depth.tobytes() == odepth.tobytes()
    assert n > 100


def test_stereo_matches_on_an_image_taller_than_the_row_index(oracle):
    """orbx_compute_stereo_matches takes the caller's keypoints and pyramids: an image of more than 8192 rows (more than the row index has
    buckets: two rows share a bucket, row_shift = 1) with synthetic keypoints -- Hamming stage, SAD stage and rejection == oracle."""
    import orb_slam3_amd as osa
    from orb_slam3_amd.extractor import KP_DTYPE
    rng = np.random.default_rng(11)
    w, h, n = 64, 9000, 600
    sf = np.array([1.0, 1.2], np.float32)
    isf = (1.0 / sf).astype(np.float32)
    pyl = [rng.integers(0, 256, (h, w), dtype=np.uint8), rng.integers(0, 256, (7500, 53), dtype=np.uint8)]
    # the right images: the left ones shifted by the keypoints' disparity plus noise (a noise-free copy would give SAD 0 everywhere, median 0,
    # and the reference's `dist < 1.5 * 1.4 * median` would reject every match)
    noisy = lambda a: np.clip(a.astype(np.int32) + rng.integers(-12, 13, a.shape), 0, 255).astype(np.uint8)
    pyr = [noisy(np.roll(pyl[0], -3, axis=1)), noisy(np.roll(pyl[1], -2, axis=1))]
    kl = np.zeros(n, KP_DTYPE)
    kl["x"] = rng.uniform(30, 44, n).astype(np.float32)
    kl["y"] = rng.uniform(20, h - 20, n).astype(np.float32)
    kl["octave"] = rng.integers(0, 2, n)
    kl["size"] = 31.0
    kl["class_id"] = -1
    dl = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    kr = kl.copy()
    kr["x"] = kl["x"] - np.where(kl["octave"] == 0, 3.0, 2.4).astype(np.float32) + rng.uniform(-0.3, 0.3, n).astype(np.float32)
    kr["y"] = kl["y"] + rng.uniform(-1.5, 1.5, n).astype(np.float32)
    dr = dl ^ np.packbits(rng.random((n, 256)) < 0.03, axis=1, bitorder="little")
    perm = rng.permutation(n)
    kr, dr = kr[perm], np.ascontiguousarray(dr[perm])
    bf, b = 40.0, 1.0
    on, our, odepth, _, _ = oracle.compute_stereo_matches(kl, dl, kr, dr, sf, isf, pyl, pyr, bf, b)
    m = osa.ORBmatcher()
    got_n, ur, depth = m.ComputeStereoMatches(kl, dl, kr, dr, sf, isf, pyl, pyr, bf, b)
    assert got_n == on and ur.tobytes() == our.tobytes() and depth.tobytes() == odepth.tobytes()
    assert on > 100


def _frame_view(kps, desc, sf, w, h):
    import orb_slam3_amd as osa
    return osa.FrameView(kps, desc, 0.0, float(w), 0.0, float(h), sf)


def test_search_by_projection_frame(oracle, canvas1):
    """M2 (ORBmatcher.cc:1676-1887): consecutive EuRoC-shaped frames, th=15, with taken-mask and rotation filter."""
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    ex = osa.ORBextractor(1000, 1.2, 8, 20, 7)
    f0 = synth.frame_from_canvas(canvas1, 0, 752, 480, 1000)
    f1 = synth.frame_from_canvas(canvas1, 1, 752, 480, 1001)
    _, k0, d0 = ex(f0, None, (0, 1000))
    _, k1, d1 = ex(f1, None, (0, 1000))
    sf = ex.GetScaleFactors()
    rng = np.random.default_rng(5)
    for mode, check_ori, th in ((0, True, 15.0), (1, True, 7.0), (2, False, 15.0), (0, True, 30.0)):
        q = dict(u=k0["x"] - 2.0, v=k0["y"] - 1.0, ur=np.zeros(len(k0), np.float32), octave=k0["octave"], angle=k0["angle"],
                 desc=d0, has_obs=(rng.random(len(k0)) < 0.9).astype(np.uint8))
        occ = (rng.random(len(k1)) < 0.05).astype(np.uint8)
        grid = oracle.OracleGrid(k1, 0.0, 752.0, 0.0, 480.0)
        on, ocm = oracle.search_by_projection_frame(grid, d1, sf, q, th, mode, check_ori, None, occ)
        m = osa.ORBmatcher(0.9, check_ori)
        n, cm = m.SearchByProjectionFrame(_frame_view(k1, d1, sf, 752, 480), q, th, mode, occ)
        assert n == on and np.array_equal(cm, ocm), (mode, n, on)
        assert n > 50
        # the call's 15 input arrays travel as ONE run of its device arena (+ the two problem records behind them), match vector and count come back
        # together: the DMA submissions of a call are part of its latency (round 5: 14 + 2 -> at most 2 + 1)
        t = m.last_transfers()
        assert 1 <= t["uploads"] <= 2 and t["downloads"] == 1, t
        assert t["upload_bytes"] >= 60 * len(k1) + 60 * len(k0) and t["download_bytes"] >= 4 * len(k1) + 4, t


def test_search_by_projection_mappoints(oracle):
    """BASELINE config 4 (TUM-VI shape): SearchByProjection against 10k map-point descriptors (M1)."""
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    canvas = synth.make_canvas(4)
    ex = osa.ORBextractor(1500, 1.2, 8, 20, 7)
    rng = np.random.default_rng(9)
    kp_all, d_all = [], []
    for t in range(1, 8):
        _, k, d = ex(synth.frame_from_canvas(canvas, t, 1024, 1024, 5000 + t), None, (0, 1000))
        k = k.copy()
        k["x"] += 2.0 * t   # frame t is the canvas shifted by (2t, t): bring to frame-0 coordinates
        k["y"] += 1.0 * t
        kp_all.append(k)
        d_all.append(d)
    _, kf, df = ex(synth.frame_from_canvas(canvas, 0, 1024, 1024, 5000), None, (0, 1000))
    sf = ex.GetScaleFactors()
    src_k = np.concatenate(kp_all)[:10000]
    src_d = np.concatenate(d_all)[:10000]
    n_mp = len(src_k)
    assert n_mp == 10000
    mp = dict(proj_x=src_k["x"] + rng.normal(0, 2, n_mp).astype(np.float32),
              proj_y=src_k["y"] + rng.normal(0, 2, n_mp).astype(np.float32),
              proj_xr=np.zeros(n_mp, np.float32), level=src_k["octave"],
              view_cos=rng.uniform(0.9, 1.0, n_mp).astype(np.float32), desc=_noisy_copy(rng, src_d, 0.04),
              in_view=(rng.random(n_mp) < 0.95).astype(np.uint8), has_obs=(rng.random(n_mp) < 0.97).astype(np.uint8))
    occ = (rng.random(len(kf)) < 0.1).astype(np.uint8)
    for th in (1.0, 3.0):
        grid = oracle.OracleGrid(kf, 0.0, 1024.0, 0.0, 1024.0)
        on, ofm = oracle.search_by_projection_mappoints(grid, df, sf, mp, th, 0.8, None, occ)
        m = osa.ORBmatcher(0.8, True)
        n, fm = m.SearchByProjection(_frame_view(kf, df, sf, 1024, 1024), mp, th, occ)
        assert n == on and np.array_equal(fm, ofm), (th, n, on)
        assert n > 100


def _synthetic_frame(rng, n, w, h):
    import orb_slam3_amd as osa
    k = np.zeros(n, osa.KP_DTYPE)
    k["octave"] = rng.integers(0, 8, n)
    sc = (1.2 ** k["octave"]).astype(np.float32)
    k["x"] = (rng.uniform(20, w - 20, n) / sc).round().astype(np.float32) * sc   # level coordinates times the level's scale, as the extractor delivers them
    k["y"] = (rng.uniform(20, h - 20, n) / sc).round().astype(np.float32) * sc
    k["size"] = 31.0 * sc
    k["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    k["response"] = rng.integers(7, 200, n).astype(np.float32)
    k["class_id"] = -1
    return k, _rand_desc(rng, n)


def test_projection_matchers_at_the_largest_frame(oracle):
    """ORBX_MAX_FRAME_FEATURES (16 000) features in the current frame: the replay of the query loop keeps 10 bytes per feature in LDS (160 KB: all of a
    CU's), the grid's cells hold up to hundreds of features (candidate lists cut short, re-scans) -- M1 and M2 == oracle; one feature more is refused
    before anything is enqueued."""
    import orb_slam3_amd as osa
    rng = np.random.default_rng(21)
    W, H, N = 1024, 1024, 16000
    kf, df = _synthetic_frame(rng, N, W, H)
    sf = np.array([1.2 ** i for i in range(8)], np.float32)
    n_mp = 3000
    src = rng.choice(N, n_mp, replace=False)
    mp = dict(proj_x=kf["x"][src] + rng.normal(0, 1.5, n_mp).astype(np.float32), proj_y=kf["y"][src] + rng.normal(0, 1.5, n_mp).astype(np.float32),
              proj_xr=np.zeros(n_mp, np.float32), level=kf["octave"][src], view_cos=rng.uniform(0.9, 1.0, n_mp).astype(np.float32),
              desc=_noisy_copy(rng, df[src], 0.05), in_view=(rng.random(n_mp) < 0.95).astype(np.uint8), has_obs=(rng.random(n_mp) < 0.97).astype(np.uint8))
    occ = (rng.random(N) < 0.1).astype(np.uint8)
    grid = oracle.OracleGrid(kf, 0.0, float(W), 0.0, float(H))
    m = osa.ORBmatcher(0.8, True)
    on, ofm = oracle.search_by_projection_mappoints(grid, df, sf, mp, 3.0, 0.8, None, occ)
    n, fm = m.SearchByProjection(_frame_view(kf, df, sf, W, H), mp, 3.0, occ)
    assert n == on and np.array_equal(fm, ofm) and n > 500, (n, on)
    q = dict(u=kf["x"][src] + 1.0, v=kf["y"][src] - 1.0, ur=np.zeros(n_mp, np.float32), octave=kf["octave"][src], angle=kf["angle"][src],
             desc=_noisy_copy(rng, df[src], 0.05), has_obs=(rng.random(n_mp) < 0.9).astype(np.uint8))
    on, ocm = oracle.search_by_projection_frame(grid, df, sf, q, 15.0, 0, True, None, occ)
    n, cm = m.SearchByProjectionFrame(_frame_view(kf, df, sf, W, H), q, 15.0, 0, occ)
    assert n == on and np.array_equal(cm, ocm) and n > 500, (n, on)
    k1, d1 = _synthetic_frame(rng, N + 1, W, H)
    with pytest.raises(osa.OrbxError):
        m.SearchByProjectionFrame(_frame_view(k1, d1, sf, W, H), q, 15.0, 0, None)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_projection_matchers_under_heavy_contention(oracle, seed):
    """The replay of the sequential query loop (k_resolve_wide_t) where it is hardest: thousands of queries compete for a few hundred features packed into a
    small region, windows hold tens of candidates (lists cut short -> re-scans, several per round), the queries are noisy copies of FEW descriptors (long
    chains of queries that want the same features in the same order), some queries leave the taken-mask untouched (has_obs = 0).  M1 and M2 == oracle."""
    import orb_slam3_amd as osa
    rng = np.random.default_rng(100 + seed)
    W, H, N = 752, 480, 300 + 100 * seed
    kf, _ = _synthetic_frame(rng, N, W, H)
    kf["x"] = rng.uniform(300, 380, N).astype(np.float32)
    kf["y"] = rng.uniform(200, 260, N).astype(np.float32)
    kf["octave"] = rng.integers(0, 3, N)
    protos = _rand_desc(rng, 6)
    df = _noisy_copy(rng, protos[rng.integers(0, 6, N)], 0.08)
    sf = np.array([1.2 ** i for i in range(8)], np.float32)
    n_mp = 2500
    src = rng.integers(0, N, n_mp)
    mp = dict(proj_x=kf["x"][src] + rng.normal(0, 3, n_mp).astype(np.float32), proj_y=kf["y"][src] + rng.normal(0, 3, n_mp).astype(np.float32),
              proj_xr=np.zeros(n_mp, np.float32), level=kf["octave"][src], view_cos=rng.uniform(0.9, 1.0, n_mp).astype(np.float32),
              desc=_noisy_copy(rng, df[src], 0.05), in_view=(rng.random(n_mp) < 0.97).astype(np.uint8), has_obs=(rng.random(n_mp) < 0.8).astype(np.uint8))
    occ = (rng.random(N) < 0.05).astype(np.uint8)
    grid = oracle.OracleGrid(kf, 0.0, float(W), 0.0, float(H))
    m = osa.ORBmatcher(0.9, True)
    for th in (3.0, 8.0):
        on, ofm = oracle.search_by_projection_mappoints(grid, df, sf, mp, th, 0.9, None, occ)
        n, fm = m.SearchByProjection(_frame_view(kf, df, sf, W, H), mp, th, occ)
        assert n == on and np.array_equal(fm, ofm), (th, n, on)
    assert n > N // 4
    q = dict(u=kf["x"][src] + 1.0, v=kf["y"][src] - 1.0, ur=np.zeros(n_mp, np.float32), octave=kf["octave"][src], angle=kf["angle"][src],
             desc=_noisy_copy(rng, df[src], 0.05), has_obs=(rng.random(n_mp) < 0.8).astype(np.uint8))
    for th, mode in ((15.0, 0), (7.0, 1)):
        on, ocm = oracle.search_by_projection_frame(grid, df, sf, q, th, mode, True, None, occ)
        n, cm = m.SearchByProjectionFrame(_frame_view(kf, df, sf, W, H), q, th, mode, occ)
        assert n == on and np.array_equal(cm, ocm), (th, mode, n, on)
    assert n > N // 8


def test_match_consecutive_device_equals_host_api(oracle, canvas1):
    """The batched device-resident frame-to-frame matcher returns what M2 returns frame by frame."""
    import torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    ex = osa.ORBextractor(1000, 1.2, 8, 20, 7)
    nfr = 4
    frames = np.stack([synth.frame_from_canvas(canvas1, t, 752, 480, 1000 + t) for t in range(nfr)])
    d = torch.from_numpy(frames).cuda()
    ex.extract_batch_device(d.data_ptr(), nfr, 752, 480, 752, 752 * 480, (0, 1000))
    cap = ex.batch_view().cap
    d_match = torch.full((nfr, cap), -7, dtype=torch.int32, device="cuda")
    d_nm = torch.zeros(nfr, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()   # torch fills on ITS stream; the library's non-blocking streams do not wait for it
    ex.match_consecutive_device(d_match.data_ptr(), d_nm.data_ptr(), th=15.0, du=-2.0, dv=-1.0, check_orientation=True)
    ex.sync()
    match, nm = d_match.cpu().numpy(), d_nm.cpu().numpy()
    sf = ex.GetScaleFactors()
    outs = [ex.download(t) for t in range(nfr)]
    for t in range(1, nfr):
        _, k0, d0 = outs[t - 1]
        _, k1, d1 = outs[t]
        q = dict(u=k0["x"] - 2.0, v=k0["y"] - 1.0, ur=np.zeros(len(k0), np.float32), octave=k0["octave"], angle=k0["angle"],
                 desc=d0, has_obs=np.ones(len(k0), np.uint8))
        grid = oracle.OracleGrid(k1, 0.0, 752.0, 0.0, 480.0)
        on, ocm = oracle.search_by_projection_frame(grid, d1, sf, q, 15.0, 0, True, None, None)
        assert nm[t] == on and np.array_equal(match[t, :len(k1)], ocm), t
        assert on > 300


def test_async_download_pipeline_equals_sync(canvas1):
    """Pipelined batches (copy stream overlapping the next batch) deliver exactly the synchronous results."""
    import torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    nfr = 3
    ex = osa.ORBextractor(1000, 1.2, 8, 20, 7)
    batches = [np.stack([synth.frame_from_canvas(canvas1, 10 * b + t, 752, 480, 7000 + 10 * b + t) for t in range(nfr)]) for b in range(3)]
    d_batches = [torch.from_numpy(x).cuda() for x in batches]
    cap = ex.output_capacity(752, 480)

    def host_set():
        return dict(kps=torch.zeros((nfr, cap, 28), dtype=torch.uint8).pin_memory(), desc=torch.zeros((nfr, cap, 32), dtype=torch.uint8).pin_memory(),
                    cnt=torch.zeros(nfr, dtype=torch.int32).pin_memory(), mono=torch.zeros(nfr, dtype=torch.int32).pin_memory(),
                    match=torch.zeros((nfr, cap), dtype=torch.int32).pin_memory(), nm=torch.zeros(nfr, dtype=torch.int32).pin_memory())
    sets = [host_set() for _ in range(3)]
    for b in range(3):   # enqueue all three batches back to back, never waiting in between
        ex.extract_batch_device(d_batches[b].data_ptr(), nfr, 752, 480, 752, 752 * 480, (0, 1000))
        ex.match_consecutive_device(th=15.0, du=-2.0, dv=-1.0, check_orientation=True)
        hs = sets[b]
        if b > 0:
            ex.download_wait()
        ex.download_async(hs["kps"].data_ptr(), hs["desc"].data_ptr(), hs["cnt"].data_ptr(), hs["mono"].data_ptr(),
                          hs["match"].data_ptr(), hs["nm"].data_ptr())
    ex.download_wait()
    ex2 = osa.ORBextractor(1000, 1.2, 8, 20, 7)
    for b in range(3):
        ex2.extract_batch_device(d_batches[b].data_ptr(), nfr, 752, 480, 752, 752 * 480, (0, 1000))
        d_match = torch.full((nfr, cap), -1, dtype=torch.int32, device="cuda")
        d_nm = torch.zeros(nfr, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ex2.match_consecutive_device(d_match.data_ptr(), d_nm.data_ptr(), th=15.0, du=-2.0, dv=-1.0, check_orientation=True)
        ex2.sync()
        hs = sets[b]
        for t in range(nfr):
            mono, kps, desc = ex2.download(t)
            n = len(kps)
            assert hs["cnt"][t] == n and hs["mono"][t] == mono
            assert hs["kps"][t, :n].numpy().tobytes() == kps.tobytes()
            assert np.array_equal(hs["desc"][t, :n].numpy(), desc)
            if t > 0:
                assert hs["nm"][t] == d_nm[t].item()
                assert np.array_equal(hs["match"][t, :n].numpy(), d_match[t, :n].cpu().numpy())


# ---------------------------------------------------------------------------------------------------------
# M3 - M7
# ---------------------------------------------------------------------------------------------------------
def _two_frames(canvas, nf=1000, t0=0, t1=1, w=752, h=480):
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    ex = osa.ORBextractor(nf, 1.2, 8, 20, 7)
    _, k0, d0 = ex(synth.frame_from_canvas(canvas, t0, w, h, 1000 + t0), None, (0, 1000))
    _, k1, d1 = ex(synth.frame_from_canvas(canvas, t1, w, h, 1000 + t1), None, (0, 1000))
    return ex, k0, d0, k1, d1


def test_search_by_projection_window_reloc_and_sim3(oracle, canvas1):
    """M3 (ORBmatcher.cc:1889-2010: levels [l-1,l+1], ORBdist 100 / 64, rotation check) and M4 (ORBmatcher.cc:427-646:
    levels [l-1,l], TH_LOW*ratioHamming, KeyFrame::GetFeaturesInArea + octave gate)."""
    import orb_slam3_amd as osa
    ex, k0, d0, k1, d1 = _two_frames(canvas1)
    sf = ex.GetScaleFactors()
    rng = np.random.default_rng(21)
    F = _frame_view(k1, d1, sf, 752, 480)
    grid = oracle.OracleGrid(k1, 0.0, 752.0, 0.0, 480.0)
    lvl = k0["octave"]
    base = dict(x=k0["x"] - 2.0 + rng.normal(0, 1.0, len(k0)).astype(np.float32), y=k0["y"] - 1.0, angle=k0["angle"], desc=d0)
    occ = (rng.random(len(k1)) < 0.1).astype(np.uint8)
    for th, orbdist in ((10.0, 100.0), (3.0, 64.0)):   # Tracking.cc:3726,3740
        q = dict(base, r=(th * sf[lvl]).astype(np.float32), min_level=lvl - 1, max_level=lvl + 1)
        on, om = oracle.search_by_projection_window(grid, d1, q, orbdist, True, False, occ)
        n, mt = osa.ORBmatcher(0.9, True).SearchByProjectionWindow(F, q, orbdist, True, occ)
        assert n == on and np.array_equal(mt, om), (th, n, on)
        assert n > 100
    for th, ratio in ((8, 1.5), (5, 1.0), (3, 1.5)):    # LoopClosing.cc:755,777,964
        q = dict(base, r=(np.float32(th) * sf[lvl]).astype(np.float32), min_level=lvl - 1, max_level=lvl)
        on, om = oracle.search_by_projection_window(grid, d1, q, 50 * ratio, False, True, occ)
        n, mt = osa.ORBmatcher(0.75, True).SearchByProjectionWindow(F, q, 50 * ratio, False, occ)
        assert n == on and np.array_equal(mt, om), (th, n, on)
        assert n > 100


def test_search_for_initialization(oracle, canvas1):
    """M6 (ORBmatcher.cc:648-763): 5x extractor, level-0 keypoints, 100-px window, vMatchedDistance reassignment."""
    import orb_slam3_amd as osa
    ex, k0, d0, k1, d1 = _two_frames(canvas1, nf=5000, t0=0, t1=4)
    sf = ex.GetScaleFactors()
    for ratio, ori in ((0.9, True), (0.7, False)):
        prev_a = np.ascontiguousarray(np.stack([k0["x"], k0["y"]], axis=1).astype(np.float32))
        prev_b = prev_a.copy()
        grid = oracle.OracleGrid(k1, 0.0, 752.0, 0.0, 480.0)
        on, om = oracle.search_for_initialization(k0, d0, grid, d1, prev_a, 100, ratio, ori)
        n, m12 = osa.ORBmatcher(ratio, ori).SearchForInitialization(k0, d0, _frame_view(k1, d1, sf, 752, 480), prev_b, 100)
        assert n == on and np.array_equal(m12, om) and prev_a.tobytes() == prev_b.tobytes()
        assert n > 200 and (m12[k0["octave"] > 0] == -1).all()


def test_search_for_initialization_under_contention(oracle, canvas1):
    """M6 where the loop's state matters: descriptors drawn from 40 prototypes (tests/test_oracle_matchers_vs_reference.py::contended_init_case, pinned
    against the compiled reference there) -- dozens of candidates within TH_LOW per query, matches taken over again and again, vMatchedDistance
    skips, distance ties, candidate lists that run dry (the whole-wave re-scan of k_replay_init_lists)."""
    import orb_slam3_amd as osa
    from test_oracle_matchers_vs_reference import contended_init_case
    ex, k0, d0, k1, d1 = _two_frames(canvas1, nf=5000, t0=0, t1=4)
    sf = ex.GetScaleFactors()
    k0, d0c, k1, d1c = contended_init_case({5000: (k0, d0, k1, d1, None)})
    rescans = []
    for ratio, ori, win in ((0.9, True, 100), (1.0, False, 60), (0.95, True, 25), (0.9, True, 400)):
        prev_a = np.ascontiguousarray(np.stack([k0["x"], k0["y"]], axis=1).astype(np.float32))
        prev_b = prev_a.copy()
        grid = oracle.OracleGrid(k1, 0.0, 752.0, 0.0, 480.0)
        on, om = oracle.search_for_initialization(k0, d0c, grid, d1c, prev_a, win, ratio, ori)
        mm = osa.ORBmatcher(ratio, ori)
        n, m12 = mm.SearchForInitialization(k0, d0c, _frame_view(k1, d1c, sf, 752, 480), prev_b, win)
        assert n == on and np.array_equal(m12, om) and prev_a.tobytes() == prev_b.tobytes(), (ratio, ori, win, n, on)
        assert n > 20
        st = mm.last_replay_stats()
        rescans.append(st["rescans"])
        assert st["queries"] == int((k0["octave"] == 0).sum()) and st["rounds"] >= (st["queries"] + 63) // 64
    assert max(rescans) > 10, rescans   # the case does drive lists dry (the re-scan path is exercised, not just present)


def _bow_nodes(rng, k_a, k_b, d_a, d_b, n_nodes=100, noise=0.15):
    """Synthetic vocabulary nodes: a feature's node is a hash of its position cell, so that true correspondences
    (frame b is frame a shifted by a few pixels) mostly share a node; `noise` of them are scrambled."""
    def node(k):
        return ((np.floor(k["x"] / 60).astype(np.int64) * 7 + np.floor(k["y"] / 60).astype(np.int64) * 13 + k["octave"] * 31) % n_nodes)
    na, nb = node(k_a), node(k_b)
    flip = rng.random(len(nb)) < noise
    nb[flip] = rng.integers(0, n_nodes, flip.sum())
    return na, nb


def test_search_by_bow(oracle, canvas1):
    """M5 (ORBmatcher.cc:223-425 and 765-905) on synthetic feature vectors."""
    import orb_slam3_amd as osa
    ex, k0, d0, k1, d1 = _two_frames(canvas1)
    rng = np.random.default_rng(31)
    na, nb = _bow_nodes(rng, k0, k1, d0, d1)
    fva, fvb = osa.FeatureVector.from_node_of_feature(na), osa.FeatureVector.from_node_of_feature(nb)
    valid0 = (rng.random(len(k0)) < 0.7).astype(np.uint8)
    valid1 = (rng.random(len(k1)) < 0.8).astype(np.uint8)
    for ratio, ori in ((0.7, True), (0.9, True), (0.75, False)):
        on, om = oracle.search_by_bow_frame(d0, k0["angle"], valid0, fva, d1, k1["angle"], fvb, ratio, ori)
        n, fm = osa.ORBmatcher(ratio, ori).SearchByBoWFrame(d0, k0["angle"], valid0, fva, d1, k1["angle"], fvb)
        assert n == on and np.array_equal(fm, om), (ratio, n, on)
        assert n > 50
        on, om = oracle.search_by_bow_keyframes(d0, k0["angle"], valid0, fva, d1, k1["angle"], valid1, fvb, ratio, ori)
        n, m12 = osa.ORBmatcher(ratio, ori).SearchByBoWKeyFrames(d0, k0["angle"], valid0, fva, d1, k1["angle"], valid1, fvb)
        assert n == on and np.array_equal(m12, om), (ratio, n, on)
        assert n > 30


def test_search_for_triangulation(oracle, canvas1):
    """M7 (ORBmatcher.cc:907-1146): later equal-distance candidate wins, lazily evaluated geometric predicate."""
    import orb_slam3_amd as osa
    ex, k0, d0, k1, d1 = _two_frames(canvas1)
    rng = np.random.default_rng(41)
    na, nb = _bow_nodes(rng, k0, k1, d0, d1, n_nodes=60)
    fva, fvb = osa.FeatureVector.from_node_of_feature(na), osa.FeatureVector.from_node_of_feature(nb)
    skip0 = (rng.random(len(k0)) < 0.4).astype(np.uint8)
    skip1 = (rng.random(len(k1)) < 0.4).astype(np.uint8)
    # inject exact duplicates in frame 1 so that distance ties occur inside a node ("later wins")
    d1 = d1.copy()
    for i in range(0, len(d1) - 1, 7):
        if nb[i] == nb[i + 1]:
            d1[i + 1] = d1[i]

    def pair_ok(i, j):   # stand-in for the epipole gate + epipolarConstrain: a pure function of the pair
        return abs((k0["y"][i] - 1.0) - k1["y"][j]) < 3.0 * (1 + k0["octave"][i])

    for ori, pred in ((False, pair_ok), (True, pair_ok), (False, None), (True, None)):  # None: bCoarse, whole loop on the device
        on, om = oracle.search_for_triangulation(d0, k0["angle"], skip0, fva, d1, k1["angle"], skip1, fvb, ori, pred)
        n, m12 = osa.ORBmatcher(0.6, ori).SearchForTriangulation(d0, k0["angle"], skip0, fva, d1, k1["angle"], skip1, fvb, pred)
        assert n == on and np.array_equal(m12, om), (ori, n, on)
        assert n > 30


def test_stereo_batch_fully_on_device(oracle):
    """Frame::ComputeStereoMatches with Hamming stage, SAD refinement and median rejection all on the device, on a batch
    of KITTI-shaped rectified pairs; mvuRight / mvDepth bit-exact vs the oracle."""
    import torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    w, h, nf, nb = 1241, 376, 2000, 3
    canvas = synth.make_canvas(3, size=2600, n_shapes=4000)
    pairs = [synth.make_stereo_pair(3, t, w, h, canvas) for t in range(nb)]
    left = np.stack([p[0] for p in pairs])
    right = np.stack([p[1] for p in pairs])
    dl, dr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    exl, exr = osa.ORBextractor(nf, 1.2, 8, 20, 7), osa.ORBextractor(nf, 1.2, 8, 20, 7)
    bf, b = np.float32(0.53716 * 718.856), np.float32(0.53716)
    for rep in range(2):   # second round exercises the cross-extractor event dependencies
        exl.extract_batch_device(dl.data_ptr(), nb, w, h, w, w * h, (0, 0))
        exr.extract_batch_device(dr.data_ptr(), nb, w, h, w, w * h, (0, 0))
        exl.stereo_batch_device(exr, float(bf), float(b))
    sf, isf = exl.GetScaleFactors(), exl.GetInverseScaleFactors()
    oexl, oexr = oracle.OracleExtractor(nf, 1.2, 8, 20, 7), oracle.OracleExtractor(nf, 1.2, 8, 20, 7)
    for t in range(nb):
        nm, ur, depth = exl.stereo_download(t)
        _, kl, dsl = oexl.extract(left[t], lap=(0, 0))
        _, kr, dsr = oexr.extract(right[t], lap=(0, 0))
        pyl = [np.ascontiguousarray(oexl.level_padded(l)[19:-19, 19:-19]) for l in range(8)]
        pyr = [np.ascontiguousarray(oexr.level_padded(l)[19:-19, 19:-19]) for l in range(8)]
        on, our, odepth, obi, obd = oracle.compute_stereo_matches(kl, dsl, kr, dsr, sf, isf, pyl, pyr, float(bf), float(b))
        assert nm == on and len(ur) == len(our), (t, nm, on)
        assert ur.tobytes() == our.tobytes() and depth.tobytes() == odepth.tobytes(), t
        assert nm > 100


def _random_vocabulary(rng, k, L, ragged=True):
    """Random k-ary vocabulary of depth L (some nodes get fewer children, some leaves sit above depth L)."""
    nwords = 0
    # simple explicit construction: nodes numbered in BFS order
    nodes = [dict(children=[], depth=0)]
    q = [0]
    while q:
        i = q.pop(0)
        d = nodes[i]["depth"]
        if d == L or (ragged and d >= 2 and rng.random() < 0.1):
            continue
        nc = k if not ragged else int(rng.integers(max(2, k - 3), k + 1))
        for _ in range(nc):
            nodes.append(dict(children=[], depth=d + 1))
            nodes[i]["children"].append(len(nodes) - 1)
            q.append(len(nodes) - 1)
    n = len(nodes)
    node_desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    word_id = np.full(n, -1, np.int32)
    cp, ci = [0], []
    for i, nd in enumerate(nodes):
        if rng.random() < 0.2:
            rng.shuffle(nd["children"])       # children order is not index order in general
        ci.extend(nd["children"])
        cp.append(len(ci))
        if not nd["children"]:
            word_id[i] = nwords
            nwords += 1
    # duplicate some sibling descriptors so that distance ties occur (first child in list order must win)
    for i, nd in enumerate(nodes):
        if len(nd["children"]) >= 2 and rng.random() < 0.15:
            node_desc[nd["children"][1]] = node_desc[nd["children"][0]]
    return np.array(cp, np.int32), np.array(ci, np.int32), node_desc, word_id


def test_bow_transform(oracle, canvas1):
    """DBoW2 TemplatedVocabulary::transform (k-ary tree descent) on a random vocabulary, incl. ties and ragged trees."""
    import orb_slam3_amd as osa
    rng = np.random.default_rng(51)
    ex, k0, d0, k1, d1 = _two_frames(canvas1)
    m = osa.ORBmatcher()
    for k, L in ((10, 4), (6, 6), (17, 3)):
        cp, ci, nd, wi = _random_vocabulary(rng, k, L)
        voc = osa.ORBVocabulary(L, cp, ci, nd, wi)
        feats = np.concatenate([d0, nd[rng.integers(1, len(nd), 300)]])   # real descriptors + exact node descriptors
        for levelsup in (4, 2, 0, 9):
            w, n_ = m.BowTransform(voc, feats, levelsup)
            ow, on = oracle.bow_transform(cp, ci, nd, wi, L, levelsup, feats)
            assert np.array_equal(w, ow) and np.array_equal(n_, on), (k, L, levelsup)
        assert (w >= 0).all() and len(np.unique(w)) > 50


def test_distinctive_descriptors(oracle):
    """MapPoint::ComputeDistinctiveDescriptors batched: least-median descriptor per observation set (first minimum wins)."""
    import orb_slam3_amd as osa
    rng = np.random.default_rng(61)
    sizes = list(rng.integers(1, 40, 300)) + [0, 1, 2, 64, 65, 130, 300]
    set_ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    base = rng.integers(0, 256, (len(sizes), 32), dtype=np.uint8)
    desc = np.concatenate([_noisy_copy(rng, np.repeat(base[i:i + 1], n, axis=0), 0.08) for i, n in enumerate(sizes) if n > 0])
    desc[set_ptr[5]:set_ptr[5] + 2] = desc[set_ptr[5]]   # identical rows -> equal medians -> first wins
    got = osa.ORBmatcher().DistinctiveDescriptors(desc, set_ptr)
    want = oracle.distinctive_descriptors(desc, set_ptr)
    assert np.array_equal(got, want)
    assert got[len(sizes) - 7] == -1 and (got[:300] >= 0).all()


def test_fuse_search(oracle, canvas1):
    """Candidate loop of Fuse x2 (ORBmatcher.cc:1246-1306, 1405-1433): level gate, chi2 reprojection gate (mono 5.99 /
    stereo 7.8), first minimum wins; with and without the gate, fused and strict float evaluation."""
    import orb_slam3_amd as osa
    ex, k0, d0, k1, d1 = _two_frames(canvas1)
    sf = ex.GetScaleFactors()
    isg = ex.GetInverseScaleSigmaSquares()
    rng = np.random.default_rng(71)
    n_q = len(k0)
    lvl = k0["octave"]
    q = dict(u=k0["x"] - 2.0 + rng.normal(0, 1.2, n_q).astype(np.float32), v=k0["y"] - 1.0 + rng.normal(0, 1.2, n_q).astype(np.float32),
             ur=(k0["x"] - 20.0).astype(np.float32), r=(np.float32(3.0) * sf[lvl]).astype(np.float32), level=lvl, desc=d0)
    u_right = np.where(rng.random(len(k1)) < 0.5, k1["x"] - 18.0 + rng.normal(0, 1.0, len(k1)), -1.0).astype(np.float32)
    grid = oracle.OracleGrid(k1, 0.0, 752.0, 0.0, 480.0)
    m = osa.ORBmatcher()
    for ur, sig, strict in ((None, isg, False), (u_right, isg, False), (u_right, isg, True), (None, None, False)):
        F = osa.FrameView(k1, d1, 0.0, 752.0, 0.0, 480.0, sf, ur)
        obi, obd = oracle.fuse_search(grid, d1, ur, sig, q, fma=not strict)
        bi, bd = m.FuseSearch(F, q, sig, strict)
        assert np.array_equal(bi, obi) and np.array_equal(bd, obd), (ur is None, sig is None, strict)
        assert (bd <= 50).sum() > 100


def test_search_by_sim3(oracle, canvas1):
    """SearchBySim3 (ORBmatcher.cc:1457-1674) composed over orbx_fuse_search, against the same composition over the oracle (which
    tests/test_oracle_matchers_vs_reference.py pins to the reference's own SearchBySim3)."""
    import orb_slam3_amd as osa
    ex, k0, d0, k1, d1 = _two_frames(canvas1)
    sf = ex.GetScaleFactors()
    rng = np.random.default_rng(61)
    th = 7.5

    def side(k, d, dx, dy):
        n = len(k)
        x = (k["x"] + dx + rng.normal(0, 1.0, n)).astype(np.float32)
        valid = rng.choice([0, 1, 1, 1, 2], n).astype(np.uint8)
        return dict(valid=valid, u=x, v=(k["y"] + dy).astype(np.float32), level=k["octave"], desc=_noisy_copy(rng, d, 0.03))
    s0, s1 = side(k0, d0, -2.0, -1.0), side(k1, d1, 2.0, 1.0)
    done0 = (rng.random(len(k0)) < 0.05).astype(np.uint8)

    def one_way(s, grid, dtab, done):
        q = dict(u=s["u"], v=s["v"], ur=np.zeros(len(s["u"]), np.float32), r=(np.float32(th) * sf[s["level"]]).astype(np.float32),
                 level=s["level"], desc=s["desc"])
        bi, bd = oracle.fuse_search(grid, dtab, None, None, q, fma=True)
        ok = (s["valid"] == 1) & (bd <= 100)
        return np.where(ok if done is None else ok & (done == 0), bi, -1)
    g0, g1 = oracle.OracleGrid(k0, 0.0, 752.0, 0.0, 480.0), oracle.OracleGrid(k1, 0.0, 752.0, 0.0, 480.0)
    m1, m2 = one_way(s0, g1, d1, done0), one_way(s1, g0, d0, None)
    want = np.array([m1[i] if (m1[i] >= 0 and m2[m1[i]] == i) else -1 for i in range(len(k0))], np.int32)
    n, m12 = osa.ORBmatcher(0.75, True).SearchBySim3(_frame_view(k0, d0, sf, 752, 480), _frame_view(k1, d1, sf, 752, 480), s0, s1, th, done0)
    assert n == (want >= 0).sum() and np.array_equal(m12, want)
    assert n > 100


def test_search_for_triangulation_pinhole_device_gates(oracle, canvas1):
    """orbx_search_for_triangulation_pinhole: epipole-distance test + Pinhole::epipolarConstrain evaluated inside k_replay_bow, against
    the oracle (pinned to the reference's texts in tests/test_oracle_{matchers,frame}_vs_reference.py), FMA and strict float modes,
    with near-threshold pairs made by placing KF2 keypoints a hair off the 1.96-sigma band of their KF1 partners."""
    import orb_slam3_amd as osa
    ex, k0, d0, k1, d1 = _two_frames(canvas1)
    sf = ex.GetScaleFactors()
    sg = (sf * sf).astype(np.float32)
    rng = np.random.default_rng(43)
    na, nb = _bow_nodes(rng, k0, k1, d0, d1, 60)
    fva, fvb = osa.FeatureVector.from_node_of_feature(na), osa.FeatureVector.from_node_of_feature(nb)
    s0 = (rng.random(len(k0)) < 0.3).astype(np.uint8)
    s1 = (rng.random(len(k1)) < 0.3).astype(np.uint8)
    ur0 = np.where(rng.random(len(k0)) < 0.2, k0["x"] - 5.0, -1.0).astype(np.float32)
    ur1 = np.where(rng.random(len(k1)) < 0.2, k1["x"] - 5.0, -1.0).astype(np.float32)
    K = np.array([[458.654, 0, 367.215], [0, 457.296, 248.375], [0, 0, 1]])
    t = np.array([0.11, 0.004, 0.01])
    th = 0.01
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F = (np.linalg.inv(K).T @ tx @ R @ np.linalg.inv(K)).astype(np.float32)
    ep = (410.0, 236.0)
    m = osa.ORBmatcher(0.6, True)
    total = 0
    for ori, coarse, u0, u1, strict in ((True, False, None, None, False), (True, False, ur0, ur1, True), (False, True, None, ur1, False),
                                        (True, False, None, None, True)):
        m.mbCheckOrientation = ori
        on, om = oracle.search_for_triangulation_pinhole(k0, d0, s0, u0, fva, k1, d1, s1, u1, fvb, sf, sg, F, ep, coarse, ori, fma=not strict)
        n, m12 = m.SearchForTriangulationPinhole(k0, d0, s0, fva, k1, d1, s1, fvb, sf, sg, F, ep, u0, u1, coarse, strict)
        assert n == on and np.array_equal(m12, om), (ori, coarse, strict, n, on)
        total += n
    assert total > 100
    # gate arithmetic alone, on pairs a few ulp from the threshold: two key frames of one feature each per pair would be slow, so
    # use one query against many candidates sharing its vocabulary node and identical descriptors (distance 0 everywhere: the
    # LAST passing candidate wins, :1017), sweeping the candidates' y across the band edge
    nq = 64
    kq = np.zeros(1, k0.dtype)
    kq["x"], kq["y"], kq["octave"] = 300.0, 200.0, 0
    dq = d0[:1]
    F64 = F.astype(np.float64)
    a = 300.0 * F64[0, 0] + 200.0 * F64[1, 0] + F64[2, 0]
    b = 300.0 * F64[0, 1] + 200.0 * F64[1, 1] + F64[2, 1]
    c = 300.0 * F64[0, 2] + 200.0 * F64[1, 2] + F64[2, 2]
    for trial in range(20):
        kc = np.zeros(nq, k0.dtype)
        kc["octave"] = rng.integers(0, 8, nq)
        x2 = rng.uniform(100, 700, nq)
        d = np.sqrt(3.84 * sg[kc["octave"]].astype(np.float64)) * np.sqrt(a * a + b * b) * rng.choice([-1, 1], nq) * (1 + rng.normal(0, 2e-7, nq))
        kc["x"], kc["y"] = x2.astype(np.float32), ((d - c - a * x2) / b).astype(np.float32)
        dc = np.repeat(dq, nq, axis=0)
        fq, fc = osa.FeatureVector.from_node_of_feature(np.zeros(1, np.int64)), osa.FeatureVector.from_node_of_feature(np.zeros(nq, np.int64))
        z1, zq = np.zeros(1, np.uint8), np.zeros(nq, np.uint8)
        for strict in (False, True):
            on, om = oracle.search_for_triangulation_pinhole(kq, dq, z1, None, fq, kc, dc, zq, None, fc, sf, sg, F, (1e6, 1e6), False, False, fma=not strict)
            m.mbCheckOrientation = False
            n, m12 = m.SearchForTriangulationPinhole(kq, dq, z1, fq, kc, dc, zq, fc, sf, sg, F, (1e6, 1e6), None, None, False, strict)
            assert n == on and np.array_equal(m12, om), (trial, strict, m12, om)


def _fisheye_keyframes(rng, n_pts=420):
    from orb_slam3_amd import synth
    return synth.make_fisheye_keyframes(rng, n_pts)


def test_kb8_epipolar_gate_every_verdict_on_the_device(oracle):
    """KannalaBrandt8::epipolarConstrain evaluated on the device for 200 000 independent keypoint pairs (orbx_debug_kb8_epipolar: the device function k_tri_kb8
    calls) -- every verdict against the oracle's: true correspondences with 0.2 .. 5 px of noise (verdicts of both kinds, many near the chi-square bounds),
    unrelated pairs, all four camera pairings of a TUM-VI-like rig, all eight level variances.  Through the search only a query's winner shows; this
    compares the Newton unprojections, the glibc tanf / atan2f restatements, the double cos / sin and the JacobiSVD sweeps as the hardware executes them."""
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    rng = np.random.default_rng(4242)
    chunks = []
    for rep in range(20):
        k1, nl1, _, id1, k2, nl2, _, id2, R12, t12, cams = synth.make_fisheye_keyframes(rng, 2500)
        # true correspondences (same 3-D point) in all camera pairings + as many unrelated pairs
        pos2 = {}
        for j, p in enumerate(id2):
            pos2.setdefault(int(p), []).append(j)
        a, b = [], []
        for i, p in enumerate(id1):
            for j in pos2.get(int(p), ()):
                a.append(i); b.append(j)
        a, b = np.array(a), np.array(b)
        ra, rb_ = rng.integers(0, len(k1), len(a)), rng.integers(0, len(k2), len(a))
        a, b = np.concatenate([a, ra]), np.concatenate([b, rb_])
        sel = (2 * (a >= nl1) + (b >= nl2)).astype(np.uint8)
        sg = (np.array([1.2 ** i for i in range(8)], np.float32) ** 2).astype(np.float32)
        chunks.append((np.stack([k1["x"][a], k1["y"][a]], 1), np.stack([k2["x"][b], k2["y"][b]], 1), sg[k1["octave"][a]], sg[k2["octave"][b]], sel, R12, t12, cams))
    m = osa.ORBmatcher()
    total = ok_sum = 0
    for xy1, xy2, s1, s2, sel, R12, t12, cams in chunks:
        want = np.zeros(len(sel), np.uint8)
        for q in range(4):
            idx = np.nonzero(sel == q)[0]
            o, _ = oracle.kb8_epipolar_constrain(cams[q >> 1], cams[q & 1], xy1[idx], xy2[idx], R12[q], t12[q], s1[idx], s2[idx])
            want[idx] = o
            assert len(idx) > 500
        got = m.DebugKb8Epipolar(cams, cams, R12, t12, xy1, xy2, s1, s2, sel)
        assert np.array_equal(got, want), (int((got != want).sum()), len(want))
        total += len(want); ok_sum += int(want.sum())
    assert total > 150000 and 0.1 < ok_sum / total < 0.6, (total, ok_sum)


def test_search_for_triangulation_fisheye_every_pair_on_the_gate(oracle):
    """The same with descriptors that are all noisy copies of ONE pattern: every pair of a vocabulary node passes the distance test, so a node's pair list
    (k_tri_kb8, 2048 entries in LDS) fills and is flushed several times per query chunk, and the gate alone decides -- equal distances resolve to the later
    candidate (ORBmatcher.cc:1017)."""
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    rng = np.random.default_rng(811)
    k1, nl1, d1, id1, k2, nl2, d2, id2, R12, t12, cams = synth.make_fisheye_keyframes(rng, 170)
    proto = _rand_desc(rng, 1)
    d1 = _noisy_copy(rng, np.repeat(proto, len(k1), 0), 0.03)
    d2 = _noisy_copy(rng, np.repeat(proto, len(k2), 0), 0.03)
    sg = (np.array([1.2 ** i for i in range(8)], np.float32) ** 2).astype(np.float32)
    fv1, fv2 = osa.FeatureVector.from_node_of_feature(id1 % 2), osa.FeatureVector.from_node_of_feature(id2 % 2)   # two nodes of ~110 x ~110 features
    s1 = (rng.random(len(k1)) < 0.1).astype(np.uint8)
    s2 = (rng.random(len(k2)) < 0.1).astype(np.uint8)
    m = osa.ORBmatcher(0.6, True)
    for ori in (True, False):
        m.mbCheckOrientation = ori
        on, om = oracle.search_for_triangulation_kb8(k1, nl1, d1, s1, fv1, k2, nl2, d2, s2, fv2, sg, sg, cams, cams, R12, t12, False, ori)
        n, m12 = m.SearchForTriangulationKB8(k1, nl1, d1, s1, fv1, k2, nl2, d2, s2, fv2, sg, sg, cams, cams, R12, t12, False)
        assert n == on and np.array_equal(m12, om), (ori, n, on)
    assert on > 20


@pytest.mark.parametrize("seed", [1, 2])
def test_search_for_triangulation_fisheye_gate_on_device(oracle, seed):
    """orbx_search_for_triangulation_kb8: KannalaBrandt8::epipolarConstrain (unproject, parallax, JacobiSVD triangulation, depth, two reprojection tests) for the
    camera pair each candidate selects, evaluated by k_tri_kb8 -- against the oracle's SearchForTriangulation with the oracle's gate as its pair predicate
    (pinned to the reference's text in tests/test_oracle_geometry.py), with and without the rotation check, and bCoarse."""
    import orb_slam3_amd as osa
    rng = np.random.default_rng(700 + seed)
    k1, nl1, d1, id1, k2, nl2, d2, id2, R12, t12, cams = _fisheye_keyframes(rng)
    sg = (np.array([1.2 ** i for i in range(8)], np.float32) ** 2).astype(np.float32)
    nodes = 40 if seed == 1 else 4   # seed 2: more than 64 features per vocabulary node (the kernel's big-node form)
    fv1, fv2 = osa.FeatureVector.from_node_of_feature(id1 % nodes), osa.FeatureVector.from_node_of_feature(id2 % nodes)
    s1 = (rng.random(len(k1)) < 0.2).astype(np.uint8)
    s2 = (rng.random(len(k2)) < 0.2).astype(np.uint8)
    # the gate for every pair, by camera pair
    table = np.zeros((len(k1), len(k2)), bool)
    for r1 in (0, 1):
        i1 = np.arange(nl1) if r1 == 0 else np.arange(nl1, len(k1))
        for r2 in (0, 1):
            i2 = np.arange(nl2) if r2 == 0 else np.arange(nl2, len(k2))
            a, b = np.repeat(i1, len(i2)), np.tile(i2, len(i1))
            ok, _ = oracle.kb8_epipolar_constrain(cams[r1], cams[r2], np.stack([k1["x"][a], k1["y"][a]], 1), np.stack([k2["x"][b], k2["y"][b]], 1),
                                                  R12[2 * r1 + r2], t12[2 * r1 + r2], sg[k1["octave"][a]], sg[k2["octave"][b]])
            table[np.ix_(i1, i2)] = ok.reshape(len(i1), len(i2)).astype(bool)
    assert 0.002 < table.mean() < 0.5
    m = osa.ORBmatcher(0.6, True)
    total = 0
    for ori, coarse in ((True, False), (False, False), (True, True)):
        m.mbCheckOrientation = ori
        on, om = oracle.search_for_triangulation(d1, k1["angle"], s1, fv1, d2, k2["angle"], s2, fv2, ori, None if coarse else (lambda i, j: table[i, j]))
        n, m12 = m.SearchForTriangulationKB8(k1, nl1, d1, s1, fv1, k2, nl2, d2, s2, fv2, sg, sg, cams, cams, R12, t12, coarse)
        cn, cm = oracle.search_for_triangulation_kb8(k1, nl1, d1, s1, fv1, k2, nl2, d2, s2, fv2, sg, sg, cams, cams, R12, t12, coarse, ori)   # the lazy gate in C
        assert cn == on and np.array_equal(cm, om)
        assert n == on and np.array_equal(m12, om), (ori, coarse, n, on)
        total += n
        if not coarse:
            hit = m12 >= 0
            assert hit.sum() > 30 and (id1[hit] == id2[m12[hit]]).mean() > 0.9      # the gate lets the true correspondences through ...
            assert (m12[hit] >= nl2).sum() > 5 and (np.nonzero(hit)[0] >= nl1).sum() > 5   # ... in all four camera pairs
    assert total > 150
