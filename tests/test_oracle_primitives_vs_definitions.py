"""Independent cross-checks of the oracle's restated OpenCV primitives ([OCV-recalled]; OpenCV itself is absent from the image and
from /root/reference, so these stay "parity unpinned" in the strict sense).  Each primitive is compared, on whole images, with a
second implementation written here from the mathematical DEFINITION in vectorised numpy -- different code, different structure:
  FAST-9/16 + 3x3 NMS   exact (corner set, order, scores)
  copyMakeBorder        exact (np.pad 'reflect' == BORDER_REFLECT_101)
  BFMatcher knn k=2     exact (stable argsort of the distance matrix)
  resize INTER_LINEAR   within 1 grey level of float64 bilinear interpolation on the half-pixel-centre geometry
  GaussianBlur 7x7 s=2  exact against the rounded separable convolution with the 8-bit kernel, which is the sampled Gaussian
                        rounded to 1/256; within the resulting quantisation budget of the true Gaussian
  fastAtan2             within the documented 0.3 degrees of atan2
The +-1 bounds are what fixed-point arithmetic (Q11 coefficients / 8-bit kernels) allows; a wrong sampling geometry, tap order,
border rule or kernel would exceed them by far."""
import numpy as np

from oracle import oracle_binding as ob
from orb_slam3_amd import synth

CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1),
          (-2, 2), (-1, 3)]   # (dx, dy), Bresenham circle of radius 3 in the order of the FAST paper


def _fast_by_definition(img, t):
    """Segment test from the definition: p is a corner iff 9 contiguous circle pixels are all > p+t or all < p-t; its score is the
    largest threshold for which it still is one; non-maximum suppression keeps scores strictly above all 8 neighbours."""
    h, w = img.shape
    I = img.astype(np.int32)
    c = I[3:h - 3, 3:w - 3]
    d = np.stack([c - I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in CIRCLE])   # centre minus circle pixel
    d2 = np.concatenate([d, d[:8]])
    best = np.full(c.shape, -10 ** 6)
    for k in range(16):
        arc = d2[k:k + 9]
        best = np.maximum(best, np.maximum(arc.min(axis=0), (-arc).min(axis=0)))
    score = np.zeros((h, w), np.int32)
    score[3:h - 3, 3:w - 3] = np.where(best > t, best - 1, 0)     # corner iff min |diff| over some arc exceeds t; score = that - 1
    s = score
    keep = np.zeros_like(s, bool)
    inner = s[3:h - 3, 3:w - 3]          # OpenCV also emits corners in the first / last scored row and column: their outer
    m = inner > 0                        # neighbours are unscored pixels, i.e. score 0
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx or dy:
                m &= inner > s[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx]
    keep[3:h - 3, 3:w - 3] = m
    ys, xs = np.nonzero(keep)      # row-major = OpenCV's emission order
    return xs, ys, s[ys, xs]


def test_fast_detector_equals_definition():
    rng = np.random.default_rng(0)
    imgs = [synth.make_test_image(11, 160, 120), synth.make_test_image(12, 97, 83),
            rng.integers(0, 256, (64, 80), dtype=np.uint8), (rng.integers(0, 4, (50, 50)) * 60).astype(np.uint8)]
    total = 0
    for img in imgs:
        for t in (7, 20, 40):
            k = ob.fast9_16(img, t)
            xs, ys, sc = _fast_by_definition(img, t)
            assert len(k) == len(xs), (img.shape, t, len(k), len(xs))
            assert np.array_equal(k["x"].astype(int), xs) and np.array_equal(k["y"].astype(int), ys)
            assert np.array_equal(k["response"].astype(int), sc)
            total += len(k)
    assert total > 500


def test_border_equals_numpy_reflect():
    rng = np.random.default_rng(1)
    L = ob.lib()
    for w, h in ((40, 30), (21, 77), (64, 20)):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        padded = np.zeros((h + 38, w + 38), np.uint8)
        padded[19:-19, 19:-19] = img
        L.orbo_border_reflect101(ob._p(padded), w, h, padded.strides[0], 19)
        assert np.array_equal(padded, np.pad(img, 19, mode="reflect"))


def test_knn2_equals_stable_argsort():
    rng = np.random.default_rng(2)
    T = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    T[40] = T[200]
    Q = T[rng.integers(0, 300, 120)] ^ np.packbits(rng.random((120, 256)) < 0.08, axis=1, bitorder="little")
    idx, dist = ob.knn2(Q, T)
    D = np.unpackbits(Q[:, None, :] ^ T[None, :, :], axis=2).sum(axis=2)
    order = np.argsort(D, axis=1, kind="stable")[:, :2]
    assert np.array_equal(idx, order) and np.array_equal(dist, np.take_along_axis(D, order, axis=1))


def _bilinear_float(src, dw, dh):
    sh, sw = src.shape
    fx = (np.arange(dw) + 0.5) * (sw / dw) - 0.5
    fy = (np.arange(dh) + 0.5) * (sh / dh) - 0.5
    x0 = np.floor(fx).astype(int); ax = fx - x0
    y0 = np.floor(fy).astype(int); ay = fy - y0
    xa, xb = np.clip(x0, 0, sw - 1), np.clip(x0 + 1, 0, sw - 1)     # replicate at the image edge
    ya, yb = np.clip(y0, 0, sh - 1), np.clip(y0 + 1, 0, sh - 1)
    S = src.astype(np.float64)
    top = S[ya][:, xa] * (1 - ax) + S[ya][:, xb] * ax
    bot = S[yb][:, xa] * (1 - ax) + S[yb][:, xb] * ax
    return top * (1 - ay)[:, None] + bot * ay[:, None]


def test_resize_within_one_level_of_float_bilinear():
    img = synth.make_test_image(13, 376, 240)
    worst = 0.0
    for dw, dh in ((313, 200), (261, 167), (188, 120), (376, 240), (301, 193)):    # the 1.2 chain and odd targets
        out = ob.resize_linear(img, dw, dh).astype(np.float64)
        worst = max(worst, np.abs(out - _bilinear_float(img, dw, dh)).max())
    assert worst <= 1.0, worst


def test_blur_equals_rounded_float_convolution_with_the_8bit_kernel():
    """GaussianBlur(7x7, sigma 2) on 8U runs in fixed point with an 8-bit kernel summing to 256.  (a) that kernel is the sampled,
    normalised Gaussian rounded to 1/256 (each tap off by < 1/256); (b) the result is the exactly rounded separable convolution
    with that kernel under BORDER_REFLECT_101 (float64 arithmetic here, integer arithmetic in the oracle)."""
    img = synth.make_test_image(14, 200, 150)
    x = np.arange(7) - 3
    g = np.exp(-x * x / (2 * 2.0 * 2.0))
    g /= g.sum()                                   # getGaussianKernel(7, 2)
    P = np.pad(img.astype(np.float64), 3, mode="reflect")      # BORDER_REFLECT_101
    for ocv440, gq in ((False, [18, 34, 48, 56, 48, 34, 18]), (True, [18, 34, 49, 55, 49, 34, 18])):
        gq = np.array(gq, np.float64)
        assert gq.sum() == (257 if ocv440 else 256) and np.abs(gq - 256 * g).max() < 1.0   # <= 4.5.0 rounded every tap: 257
        tmp = sum(gq[k] * P[:, k:k + 200] for k in range(7))
        want = np.minimum(np.floor(sum(gq[k] * tmp[k:k + 150, :] for k in range(7)) / 65536 + 0.5), 255)
        assert np.array_equal(ob.gauss7(img, ocv440=ocv440).astype(np.float64), want)
    # and the blur stays within the quantisation budget of the true Gaussian: 2 passes x 255 x sum |gq/256 - g| + rounding
    tmp = sum(g[k] * P[:, k:k + 200] for k in range(7))
    true = sum(g[k] * tmp[k:k + 150, :] for k in range(7))
    budget = 2 * 255 * np.abs(np.array([18, 34, 48, 56, 48, 34, 18]) / 256 - g).sum() + 0.5
    assert np.abs(ob.gauss7(img).astype(np.float64) - true).max() <= budget


def test_fast_atan2_accuracy_dense():
    rng = np.random.default_rng(3)
    y, x = rng.normal(0, 1e4, 20000), rng.normal(0, 1e4, 20000)
    got = np.array([ob.fast_atan2(float(a), float(b)) for a, b in zip(y.astype(np.float32), x.astype(np.float32))])
    want = np.degrees(np.arctan2(y.astype(np.float32).astype(np.float64), x.astype(np.float32).astype(np.float64))) % 360
    assert np.abs((got - want + 180) % 360 - 180).max() < 0.3


def test_resize_at_exactly_two_is_the_box_average():
    """cv::resize(INTER_LINEAR) at a scale of exactly 2 in both directions is re-routed to INTER_AREA, whose fast 2 x 2 path for 8-bit images is
    (a + b + c + d + 2) >> 2 (imgproc/src/resize.cpp: "in case of scale_x && scale_y is equal to 2 INTER_AREA (fast) also is equal to
    INTER_LINEAR").  The fixed-point bilinear restatement needs no second path for it: at exactly 2 x the source coordinate is 2x + 0.5, both
    horizontal and both vertical weights are 1024 of 2048, ((1024 * ((a + b) * 1024 >> 4)) >> 16) = a + b exactly, and the result IS the box
    average.  Checked here on random images (even sizes halve exactly; ORBextractor.cc:1183 with scaleFactor == 2.0)."""
    rng = np.random.default_rng(5)
    for (h, w) in ((480, 752), (376, 1240), (2, 2), (10, 6), (768, 1024)):
        src = rng.integers(0, 256, (h, w), dtype=np.uint8)
        got = ob.resize_linear(src, w // 2, h // 2)
        s = src.astype(np.int32)
        want = ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
        assert np.array_equal(got, want), (h, w)
