"""GPU tests at the bench's shape: B = 256 frames per batch, depth-2 pipeline over >= 20 steps, ALTERNATING inputs (so that
nothing can pass by re-reading the previous batch's state), host-resident and device-resident input paths, sampled frames
and match vectors of every step compared bit-for-bit with the CPU oracle.  Also: the batched map-point projection search
(BASELINE config 4), the host-input entry point and the all-frames stereo download."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, NF = 752, 480, 1000


def _oracle_check_pair(ob, oex, sf, frames, hs, s):
    """frames s, s+1 of a delivered batch + the match vector of the pair against the oracle."""
    outs = []
    for f in (s, s + 1):
        mono, k, d = oex.extract(frames[f], lap=(0, 1000))
        n = int(hs["cnt"][f])
        assert n == len(k) and int(hs["mono"][f]) == mono, (f, n, len(k))
        assert hs["kps"][f, :n].numpy().tobytes() == k.tobytes(), f
        assert np.array_equal(hs["desc"][f, :n].numpy(), d), f
        outs.append((k, d))
    (k0, d0), (k1, d1) = outs
    q = dict(u=k0["x"] - 2.0, v=k0["y"] - 1.0, ur=np.zeros(len(k0), np.float32), octave=k0["octave"], angle=k0["angle"],
             desc=d0, has_obs=np.ones(len(k0), np.uint8))
    grid = ob.OracleGrid(k1, 0.0, float(W), 0.0, float(H))
    on, ocm = ob.search_by_projection_frame(grid, d1, sf, q, 15.0, 0, True, None, None)
    assert int(hs["nm"][s + 1]) == on and np.array_equal(hs["match"][s + 1, :len(k1)].numpy(), ocm), s
    assert on > 200


def test_bench_shape_pipeline_alternating_inputs(oracle):
    _run_pipeline(oracle, 256, 24)


@pytest.mark.parametrize("B", [256, 9])
def test_every_frame_and_match_vector_of_a_step(oracle, B):
    """One delivered step compared WHOLE: all B frames (keypoints, descriptors, mono split) and all B - 1 match vectors against the oracle
    (frame-parallel on the host cores).  The frame index drives the XCD mapping (frame = 8 * blockIdx.z + blockIdx.x for B >= 8), so sampled
    frames do not cover it; B = 9 leaves the last XCD group with a single frame."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    import torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    canvas = synth.make_canvas(10)
    frames = np.stack([synth.frame_from_canvas(canvas, t, W, H, 10000 + t) for t in range(B)])
    d_frames = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    ex = osa.ORBextractor(NF, 1.2, 8, 20, 7)
    cap = ex.output_capacity(W, H)
    hs = dict(kps=torch.zeros((B, cap, 28), dtype=torch.uint8).pin_memory(), desc=torch.zeros((B, cap, 32), dtype=torch.uint8).pin_memory(),
              cnt=torch.zeros(B, dtype=torch.int32).pin_memory(), mono=torch.zeros(B, dtype=torch.int32).pin_memory(),
              match=torch.zeros((B, cap), dtype=torch.int32).pin_memory(), nm=torch.zeros(B, dtype=torch.int32).pin_memory())
    for _ in range(2):   # the second step overlaps the first one's matcher and download, as in the streaming loop
        ex.extract_batch_device(d_frames.data_ptr(), B, W, H, W, W * H, (0, 1000))
        ex.match_consecutive_device(th=15.0, du=-2.0, dv=-1.0, check_orientation=True)
        ex.download_async(hs["kps"].data_ptr(), hs["desc"].data_ptr(), hs["cnt"].data_ptr(), hs["mono"].data_ptr(), hs["match"].data_ptr(), hs["nm"].data_ptr())
    ex.download_wait()
    ex.download_wait()
    ex.sync()
    sf = oracle.OracleExtractor(NF, 1.2, 8, 20, 7).tables()["scale"]
    nt = max(1, min(os.cpu_count() or 1, 64, B))

    def extract_chunk(fs):
        oex = oracle.OracleExtractor(NF, 1.2, 8, 20, 7)
        return [(f,) + tuple(oex.extract(frames[f], lap=(0, 1000))) for f in fs]

    with ThreadPoolExecutor(nt) as pool:
        res = {f: (mono, k, d) for chunk in pool.map(extract_chunk, [list(range(B))[i::nt] for i in range(nt)]) for f, mono, k, d in chunk}
        for f in range(B):
            mono, k, d = res[f]
            n = int(hs["cnt"][f])
            assert n == len(k) and int(hs["mono"][f]) == mono, (f, n, len(k))
            assert hs["kps"][f, :n].numpy().tobytes() == k.tobytes() and np.array_equal(hs["desc"][f, :n].numpy(), d), f

        def match_pair(s0):
            (_, k0, d0), (_, k1, d1) = res[s0], res[s0 + 1]
            q = dict(u=k0["x"] - 2.0, v=k0["y"] - 1.0, ur=np.zeros(len(k0), np.float32), octave=k0["octave"], angle=k0["angle"], desc=d0,
                     has_obs=np.ones(len(k0), np.uint8))
            on, ocm = oracle.search_by_projection_frame(oracle.OracleGrid(k1, 0.0, float(W), 0.0, float(H)), d1, sf, q, 15.0, 0, True, None, None)
            return s0, on, ocm, len(k1)

        for s0, on, ocm, n1 in pool.map(match_pair, range(B - 1)):
            assert int(hs["nm"][s0 + 1]) == on and np.array_equal(hs["match"][s0 + 1, :n1].numpy(), ocm), s0


def _run_pipeline(oracle, B, steps, canvas_size=2048, n_shapes=2400):
    import torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    canvases = [synth.make_canvas(10, size=canvas_size, n_shapes=n_shapes), synth.make_canvas(11, size=canvas_size, n_shapes=n_shapes)]
    sets = [np.stack([synth.frame_from_canvas(c, t, W, H, 1000 * (10 + i) + t) for t in range(B)]) for i, c in enumerate(canvases)]
    d_sets = [torch.from_numpy(x).cuda() for x in sets]
    h_sets = [torch.from_numpy(x).pin_memory() for x in sets]
    torch.cuda.synchronize()
    ex = osa.ORBextractor(NF, 1.2, 8, 20, 7)
    cap = ex.output_capacity(W, H)

    def host_set():
        return dict(kps=torch.zeros((B, cap, 28), dtype=torch.uint8).pin_memory(), desc=torch.zeros((B, cap, 32), dtype=torch.uint8).pin_memory(),
                    cnt=torch.zeros(B, dtype=torch.int32).pin_memory(), mono=torch.zeros(B, dtype=torch.int32).pin_memory(),
                    match=torch.zeros((B, cap), dtype=torch.int32).pin_memory(), nm=torch.zeros(B, dtype=torch.int32).pin_memory())
    host = [host_set(), host_set()]
    oex = oracle.OracleExtractor(NF, 1.2, 8, 20, 7)
    sf = oex.tables()["scale"]
    rng = np.random.default_rng(3)

    def enqueue(i):
        k, hs = i % 2, host[i % 2]          # batch i uses input set i % 2 ...
        if (i // 2) % 2:                     # ... and alternates between the host-input and the device-input entry point
            ex.extract_batch_host(h_sets[k].data_ptr(), B, W, H, W, W * H, (0, 1000))
        else:
            ex.extract_batch_device(d_sets[k].data_ptr(), B, W, H, W, W * H, (0, 1000))
        ex.match_consecutive_device(th=15.0, du=-2.0, dv=-1.0, check_orientation=True)
        ex.download_async(hs["kps"].data_ptr(), hs["desc"].data_ptr(), hs["cnt"].data_ptr(), hs["mono"].data_ptr(),
                          hs["match"].data_ptr(), hs["nm"].data_ptr())

    for i in range(steps + 1):
        if i < steps:
            enqueue(i)
        if i >= 1:
            ex.download_wait()
            j = i - 1
            # the batch in flight keeps the GPU busy while the oracle checks two random pairs of the delivered one
            for s in sorted(set(int(x) for x in rng.integers(0, B - 1, 2))) + ([0, B - 2] if j in (0, steps - 1) else []):
                _oracle_check_pair(oracle, oex, sf, sets[j % 2], host[j % 2], s)
    ex.sync()


# History of the three tests below: at the end of round 2 a child of test_pipeline_under_alternative_switches delivered 1007 keypoints
# where the oracle has 1008 -- frame 0 of batch 3, the first host-input batch into a freshly allocated input slab (canvas 11, size
# 1024, 700 shapes).  Probable cause: DevBuf::ensure cleared the slab with an asynchronous hipMemset that
# nothing ordered against the upload; the oracle gives exactly 1007 for that frame with its first 16 KB zeroed.  Fixed (ensure waits
# for the fill).  With the fix all three passed on the MI355X (profiles/r02_k_open_item_after_fix.log, 24 fresh-slab trials in the
# third test), so they are ordinary tests now.  Control run, first GPU visit of round 3 (profiles/r03_a_open_item_control.log): with the
# wait taken out again the third test fails at once (1002 instead of 1004 keypoints in frame 0 of the second slab) and the short-batch
# loop differs in 5 of 5 processes, with it 0 of 5 -- the unsynchronised clear was the cause.


def test_open_small_canvas_single_image(oracle):
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    img = synth.frame_from_canvas(synth.make_canvas(11, size=1024, n_shapes=700), 0, W, H, 11000)
    mono, kps, desc = osa.ORBextractor(NF, 1.2, 8, 20, 7)(img, None, (0, 1000))
    omono, okps, odesc = oracle.OracleExtractor(NF, 1.2, 8, 20, 7).extract(img, lap=(0, 1000))
    assert len(kps) == len(okps) == 1008 and mono == omono
    assert kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)


def test_open_small_canvas_short_batches(oracle):
    _run_pipeline(oracle, 16, 4, canvas_size=1024, n_shapes=700)


def test_open_first_host_batch_into_fresh_slab(oracle):
    """The suspected sequence itself, 12 times over: a NEW extractor, whose first call is a host-input batch (configure allocates and clears
    every buffer, the input slab is allocated, cleared and filled by the upload right away) -- first and last frame == oracle; then a
    second host batch into the other fresh slab."""
    import torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    B = 16
    canvases = [synth.make_canvas(11, size=1024, n_shapes=700), synth.make_canvas(10, size=1024, n_shapes=700)]
    sets = [np.stack([synth.frame_from_canvas(c, t, W, H, 1000 * (11 - i) + t) for t in range(B)]) for i, c in enumerate(canvases)]
    h_sets = [torch.from_numpy(x).pin_memory() for x in sets]
    oex = oracle.OracleExtractor(NF, 1.2, 8, 20, 7)
    want = [[oex.extract(sets[i][f], lap=(0, 1000)) for f in (0, B - 1)] for i in range(2)]
    assert len(want[0][0][1]) == 1008
    for rep in range(12):
        ex = osa.ORBextractor(NF, 1.2, 8, 20, 7)
        for i in range(2):   # slab 0, then slab 1
            ex.extract_batch_host(h_sets[i].data_ptr(), B, W, H, W, W * H, (0, 1000))
            for f, (omono, ok, od) in zip((0, B - 1), want[i]):
                mono, k, d = ex.download(f)
                assert len(k) == len(ok) and mono == omono and k.tobytes() == ok.tobytes() and np.array_equal(d, od), (rep, i, f, len(k), len(ok))
        del ex


def test_pipeline_under_alternative_switches():
    """The remaining switches of the library (everything on one stream; the sequential quad-tree emulation; a tiny FAST pixel queue that
    pushes most cells through the list pass) run the pipelined extract -> match -> download loop (16 frames per batch, alternating inputs
    and entry points) bit-identically to the oracle.  The switches are read once per process, hence the child processes (this file run as
    a script)."""
    import os
    import subprocess
    import sys
    for extra in ({"ORBX_OCTREE": "seq"}, {"ORBX_FAST_QCAP": "48"}):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "16", "3"], capture_output=True, text=True, env=dict(os.environ, **extra),
                           timeout=300)
        assert r.returncode == 0 and "pipeline ok" in r.stdout, (extra, r.stdout[-500:], r.stderr[-2000:])


def test_host_frames_may_be_overwritten_after_the_next_call(oracle):
    """include/orbx.h: the frames handed to orbx_extract_batch_host may be reused once the NEXT call has returned.  Two batches in flight
    (nothing waits in between); the camera thread scribbles over batch i's frames the moment call i+1 has returned -- batch i's results
    must still be those of its frames.  (Before the fix the upload of batch i could still be pending at that point.)"""
    import torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    B, steps = 48, 8
    canvas = synth.make_canvas(12)
    sets = [np.stack([synth.frame_from_canvas(canvas, 40 * i + t, W, H, 500 * i + t) for t in range(B)]) for i in range(steps)]
    bufs = [torch.zeros((B, H, W), dtype=torch.uint8).pin_memory() for _ in range(2)]
    ex = osa.ORBextractor(NF, 1.2, 8, 20, 7)
    cap = ex.output_capacity(W, H)
    host = [dict(kps=torch.zeros((B, cap, 28), dtype=torch.uint8).pin_memory(), desc=torch.zeros((B, cap, 32), dtype=torch.uint8).pin_memory(),
                 cnt=torch.zeros(B, dtype=torch.int32).pin_memory(), mono=torch.zeros(B, dtype=torch.int32).pin_memory()) for _ in range(2)]
    oex = oracle.OracleExtractor(NF, 1.2, 8, 20, 7)
    for i in range(steps + 1):
        if i < steps:
            bufs[i % 2].numpy()[:] = sets[i]
            ex.extract_batch_host(bufs[i % 2].data_ptr(), B, W, H, W, W * H, (0, 1000))
            hs = host[i % 2]
            ex.download_async(hs["kps"].data_ptr(), hs["desc"].data_ptr(), hs["cnt"].data_ptr(), hs["mono"].data_ptr(), 0, 0)
            if i >= 1:
                bufs[(i - 1) % 2].numpy()[:] = 0x5a       # call i has returned: batch i-1's frames are the caller's again
        if i >= 1:
            ex.download_wait()
            j, hs = i - 1, host[(i - 1) % 2]
            for f in (0, B // 2, B - 1):
                omono, ok, od = oex.extract(sets[j][f], lap=(0, 1000))
                n = int(hs["cnt"][f])
                assert n == len(ok) and int(hs["mono"][f]) == omono and hs["kps"][f, :n].numpy().tobytes() == ok.tobytes() and \
                    np.array_equal(hs["desc"][f, :n].numpy(), od), (j, f, n, len(ok))
    ex.sync()


def test_extract_batch_host_strided_input_equals_device_path():
    import torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    n, w, h = 5, 640, 400
    canvas = synth.make_canvas(21)
    frames = np.stack([synth.frame_from_canvas(canvas, t, w, h, 2100 + t) for t in range(n)])
    ex = osa.ORBextractor(800, 1.2, 8, 20, 7)
    ex.extract_batch_device(torch.from_numpy(frames).cuda().data_ptr(), n, w, h, w, w * h, (0, 0))
    want = [ex.download(f) for f in range(n)]
    # (a) packed pinned frames; (b) rows padded to 704 bytes; (c) padded rows and a gap between frames
    for row_stride, gap in ((w, 0), (704, 0), (704, 1000)):
        fs = row_stride * h + gap
        buf = torch.zeros(n * fs, dtype=torch.uint8).pin_memory()
        view = buf.numpy()
        for f in range(n):
            view[f * fs:f * fs + row_stride * h].reshape(h, row_stride)[:, :w] = frames[f]
        for _ in range(3):   # both input slabs get used and re-used
            ex.extract_batch_host(buf.data_ptr(), n, w, h, row_stride, fs, (0, 0))
        for f in range(n):
            mono, k, d = ex.download(f)
            assert mono == want[f][0] and k.tobytes() == want[f][1].tobytes() and np.array_equal(d, want[f][2]), (row_stride, gap, f)


def _make_mappoints(rng, kps_list, desc_list, t, n_mp):
    src = [s for s in range(max(0, t - 8), t)] or [t]
    k = np.concatenate([kps_list[s] for s in src])
    d = np.concatenate([desc_list[s] for s in src])
    idx = rng.integers(0, len(k), n_mp)
    k, d = k[idx], d[idx].copy()
    d ^= np.packbits(rng.random((n_mp, 256)) < 0.04, axis=1, bitorder="little")
    in_view = (rng.random(n_mp) < 0.95).astype(np.uint8)
    level = k["octave"].astype(np.int32)
    level[rng.random(n_mp) < 0.01] = 9        # out-of-range predicted level: skipped
    vc = rng.uniform(0.9, 1.0, n_mp).astype(np.float32)
    vc[::7] = 0.999                           # both RadiusByViewingCos branches
    return dict(proj_x=(k["x"] + rng.normal(0, 2, n_mp)).astype(np.float32), proj_y=(k["y"] + rng.normal(0, 2, n_mp)).astype(np.float32),
                proj_xr=np.zeros(n_mp, np.float32), level=level, view_cos=vc, desc=d, in_view=in_view, has_obs=np.ones(n_mp, np.uint8))


@pytest.mark.parametrize("w,h,nf,n_mp,th", [(1024, 1024, 1500, 10000, 1.0), (752, 480, 1000, 3000, 3.0)])
def test_search_mappoints_batch_device_equals_oracle(oracle, w, h, nf, n_mp, th):
    """BASELINE config 4: SearchByProjection(Frame, MapPoints) of every frame of a resident batch against 10 k map points."""
    import torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    B = 6
    canvas = synth.make_canvas(4, size=2600, n_shapes=3000)
    frames = np.stack([synth.frame_from_canvas(canvas, t, w, h, 5000 + t) for t in range(B)])
    d_frames = torch.from_numpy(frames).cuda()
    ex = osa.ORBextractor(nf, 1.2, 8, 20, 7)
    ex.extract_batch_device(d_frames.data_ptr(), B, w, h, w, w * h, (0, 1000))
    feats = [ex.download(f) for f in range(B)]
    rng = np.random.default_rng(77)
    mps = [_make_mappoints(rng, [x[1] for x in feats], [x[2] for x in feats], f, n_mp) for f in range(B)]
    dev = {k: torch.from_numpy(np.stack([m[k] for m in mps])).cuda() for k in ("proj_x", "proj_y", "level", "view_cos", "desc", "in_view")}
    cap = ex.batch_view().cap
    d_match = torch.full((B, cap), -7, dtype=torch.int32, device="cuda")
    d_nm = torch.full((B,), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()   # torch fills on ITS stream; the library's non-blocking streams do not wait for it
    for _ in range(2):   # second call = cached problem descriptors
        ex.search_mappoints_batch_device(n_mp, dev["proj_x"].data_ptr(), dev["proj_y"].data_ptr(), dev["level"].data_ptr(),
                                         dev["view_cos"].data_ptr(), dev["in_view"].data_ptr(), dev["desc"].data_ptr(), th=th, nnratio=0.8,
                                         d_match=d_match.data_ptr(), d_nmatches=d_nm.data_ptr())
    ex.sync()
    match, nm = d_match.cpu().numpy(), d_nm.cpu().numpy()
    sf = ex.GetScaleFactors()
    for f in range(B):
        _, k, d = feats[f]
        grid = oracle.OracleGrid(k, 0.0, float(w), 0.0, float(h))
        on, ofm = oracle.search_by_projection_mappoints(grid, d, sf, mps[f], th, 0.8)
        assert nm[f] == on and np.array_equal(match[f, :len(k)], ofm), f
        assert on > 300
    # one set of internal result buffers per extractor (include/orbx.h): the second batched matcher of a batch that asks for them is refused
    ex.match_consecutive_device(th=15.0, du=-2.0, dv=-1.0, check_orientation=True)
    with pytest.raises(osa.OrbxError):
        ex.search_mappoints_batch_device(n_mp, dev["proj_x"].data_ptr(), dev["proj_y"].data_ptr(), dev["level"].data_ptr(),
                                         dev["view_cos"].data_ptr(), dev["in_view"].data_ptr(), dev["desc"].data_ptr(), th=th, nnratio=0.8)
    ex.sync()


def test_stereo_download_all_equals_per_frame_download():
    import torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    B, w, h = 4, 1241, 376
    canvas = synth.make_canvas(30, size=2600, n_shapes=3000)
    pairs = [synth.make_stereo_pair(30, t, w, h, canvas) for t in range(B)]
    dl = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    dr = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    exl, exr = osa.ORBextractor(2000, 1.2, 8, 20, 7), osa.ORBextractor(2000, 1.2, 8, 20, 7)
    exl.extract_batch_device(dl.data_ptr(), B, w, h, w, w * h, (0, 0))
    exr.extract_batch_device(dr.data_ptr(), B, w, h, w, w * h, (0, 0))
    exl.stereo_batch_device(exr, 0.53716 * 718.856, 0.53716)
    cap = exl.batch_view().cap
    ur, depth, nm = np.zeros((B, cap), np.float32), np.zeros((B, cap), np.float32), np.zeros(B, np.int32)
    exl.stereo_download_all(ur.ctypes.data, depth.ctypes.data, nm.ctypes.data)
    for f in range(B):
        n1, u1, d1 = exl.stereo_download(f)
        assert nm[f] == n1 and n1 > 100
        assert ur[f, :len(u1)].tobytes() == u1.tobytes() and depth[f, :len(d1)].tobytes() == d1.tobytes()


def test_stereo_pipelined_async_downloads_equal_synchronous_ones():
    """The KITTI bench's loop shape: extraction of both cameras, orbx_stereo_batch_device, asynchronous downloads of the features and of the
    stereo result into pinned buffers, the NEXT step enqueued before the previous one is waited for (two different sets of pairs
    alternate, so that a result overwritten too early or delivered from the wrong step differs)."""
    import torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    B, w, h = 3, 620, 240
    canvas = synth.make_canvas(31, size=1400, n_shapes=1500)
    data = []
    for s0 in (0, 40):
        pairs = [synth.make_stereo_pair(31, s0 + t, w, h, canvas) for t in range(B)]
        data.append((torch.from_numpy(np.stack([p[0] for p in pairs])).cuda(), torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()))
    exl, exr = osa.ORBextractor(800, 1.2, 8, 20, 7), osa.ORBextractor(800, 1.2, 8, 20, 7)
    bf, b = 0.53716 * 718.856, 0.53716
    cap = exl.output_capacity(w, h)
    exr.output_capacity(w, h)
    pin = lambda shape, dt: torch.zeros(shape, dtype=dt).pin_memory()
    # synchronous reference results of both sets
    want = []
    for dl, dr in data:
        exl.extract_batch_device(dl.data_ptr(), B, w, h, w, w * h, (0, 0))
        exr.extract_batch_device(dr.data_ptr(), B, w, h, w, w * h, (0, 0))
        exl.stereo_batch_device(exr, bf, b)
        ur, depth, nm = np.zeros((B, cap), np.float32), np.zeros((B, cap), np.float32), np.zeros(B, np.int32)
        exl.stereo_download_all(ur.ctypes.data, depth.ctypes.data, nm.ctypes.data)
        feats = [(exl.download(f), exr.download(f)) for f in range(B)]
        assert nm.min() > 30
        want.append((ur, depth, nm, feats))
    sets = [dict(l=[pin((B, cap, 28), torch.uint8), pin((B, cap, 32), torch.uint8), pin((B,), torch.int32), pin((B,), torch.int32)],
                 r=[pin((B, cap, 28), torch.uint8), pin((B, cap, 32), torch.uint8), pin((B,), torch.int32), pin((B,), torch.int32)],
                 ur=pin((B, cap), torch.float32), depth=pin((B, cap), torch.float32), nm=pin((B,), torch.int32)) for _ in range(2)]
    steps = 5
    for i in range(steps + 1):
        if i < steps:
            dl, dr = data[i % 2]
            hs = sets[i % 2]
            exl.extract_batch_device(dl.data_ptr(), B, w, h, w, w * h, (0, 0))
            exr.extract_batch_device(dr.data_ptr(), B, w, h, w, w * h, (0, 0))
            exl.stereo_batch_device(exr, bf, b)
            exl.download_async(*[t.data_ptr() for t in hs["l"]])
            exr.download_async(*[t.data_ptr() for t in hs["r"]])
            exl.stereo_download_async(hs["ur"].data_ptr(), hs["depth"].data_ptr(), hs["nm"].data_ptr())
        if i >= 1:
            exl.download_wait(); exr.download_wait(); exl.stereo_download_wait()
            hs = sets[(i - 1) % 2]
            ur, depth, nm, feats = want[(i - 1) % 2]
            assert np.array_equal(hs["nm"].numpy(), nm)
            for f in range(B):
                for k, (mono, kps, desc) in zip("lr", feats[f]):
                    n = int(hs[k][2][f])
                    assert n == len(kps) and hs[k][0][f, :n].numpy().tobytes() == kps.tobytes() and np.array_equal(hs[k][1][f, :n].numpy(), desc), (i, f, k)
                n = int(hs["l"][2][f])
                assert hs["ur"][f, :n].numpy().tobytes() == ur[f, :n].tobytes() and hs["depth"][f, :n].numpy().tobytes() == depth[f, :n].tobytes(), (i, f)
    if not os.environ.get("ORBX_TEST_EMULATOR"):   # under the emulator every pointer is "host memory": nothing to refuse
        pageable = np.zeros((B, cap), np.float32)
        with pytest.raises(Exception):
            exl.stereo_download_async(pageable.ctypes.data, 0, 0)   # pageable memory is refused


def test_async_entry_points_refuse_pageable_host_memory():
    """orbx_extract_batch_host / orbx_batch_download_async take the caller's pointers to the copy engine: pinned memory or an error."""
    import torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    from orb_slam3_amd._lib import OrbxError
    n, w, h = 2, 320, 240
    frames = np.stack([synth.make_test_image(3 + t, w, h) for t in range(n)])
    ex = osa.ORBextractor(500, 1.2, 8, 20, 7)
    with pytest.raises(OrbxError):
        ex.extract_batch_host(frames.ctypes.data, n, w, h, w, w * h, (0, 0))          # pageable numpy memory
    pinned = torch.from_numpy(frames).pin_memory()
    ex.extract_batch_host(pinned.data_ptr(), n, w, h, w, w * h, (0, 0))
    cap = ex.batch_view().cap
    kps = np.zeros((n, cap, 28), np.uint8)
    with pytest.raises(OrbxError):
        ex.download_async(kps.ctypes.data, 0, 0, 0)
    ex.sync()
    assert len(ex.download(0)[1]) > 50


if __name__ == "__main__":   # child of test_pipeline_under_alternative_switches: python tests/test_gpu_pipeline.py <frames per batch> <steps> [small]
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from oracle import oracle_binding as _ob
    _ob.lib()
    # standard canvases by default (the switch test is about the switches); "small" = the 1024^2 / 700-shape canvases of the open item
    small = len(sys.argv) > 3 and sys.argv[3] == "small"
    _run_pipeline(_ob, int(sys.argv[1]), int(sys.argv[2]), canvas_size=1024 if small else 2048, n_shapes=700 if small else 2400)
    print("pipeline ok")
