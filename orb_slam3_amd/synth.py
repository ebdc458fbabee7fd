"""Seeded synthetic frame generator.

No dataset is available offline, so every config of BASELINE.json runs on synthetic
camera-like frames: a large canvas (smooth background + random filled convex quads of
random grey levels, i.e. plenty of FAST corners at every pyramid scale) from which frame t
is the crop at offset (2t, t) plus per-frame Gaussian sensor noise.  Pure numpy; used by
tests, bench.py and __graft_entry__.smoke().
"""
from __future__ import annotations

import numpy as np

__all__ = ["make_canvas", "make_texture_canvas", "frame_from_canvas", "make_frames", "make_stereo_pair", "make_test_image"]


def _bilinear_upsample(small: np.ndarray, size: int) -> np.ndarray:
    n = small.shape[0]
    xs = np.linspace(0, n - 1, size)
    x0 = np.clip(np.floor(xs).astype(int), 0, n - 2)
    fx = (xs - x0).astype(np.float32)
    wmat = np.zeros((size, n), np.float32)  # interpolation matrix, 2 non-zeros per row
    wmat[np.arange(size), x0] = 1 - fx
    wmat[np.arange(size), x0 + 1] += fx
    return wmat @ small.astype(np.float32) @ wmat.T


def make_canvas(seed: int, size: int = 2048, n_shapes: int = 2400) -> np.ndarray:
    """float32 canvas in [0,255]: background + `n_shapes` random filled convex quads."""
    rng = np.random.default_rng(seed)
    canvas = _bilinear_upsample(rng.uniform(40, 215, (32, 32)), size).astype(np.float32)
    for _ in range(n_shapes):
        cx, cy = rng.uniform(0, size, 2)
        half = rng.uniform(4, 48, 2)
        ang = rng.uniform(0, np.pi)
        grey = rng.uniform(0, 255)
        skew = rng.uniform(0.7, 1.3, 4)
        # 4 corners of a skewed rotated rectangle (convex)
        base = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], dtype=np.float64) * half * skew[:, None]
        c, s = np.cos(ang), np.sin(ang)
        pts = base @ np.array([[c, s], [-s, c]]) + (cx, cy)
        x0 = int(max(0, np.floor(pts[:, 0].min())))
        x1 = int(min(size, np.ceil(pts[:, 0].max()) + 1))
        y0 = int(max(0, np.floor(pts[:, 1].min())))
        y1 = int(min(size, np.ceil(pts[:, 1].max()) + 1))
        if x1 <= x0 or y1 <= y0:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1]
        inside = np.ones(xx.shape, dtype=bool)
        for k in range(4):
            ax, ay = pts[k]
            bx, by = pts[(k + 1) % 4]
            inside &= ((bx - ax) * (yy - ay) - (by - ay) * (xx - ax)) >= 0
        canvas[y0:y1, x0:x1][inside] = grey
    return canvas


def make_texture_canvas(seed: int, size: int = 2048) -> np.ndarray:
    """float32 canvas in [0,255] with NATURAL-IMAGE statistics instead of flat quads: 1/f ("pink") noise -- the amplitude spectrum of natural
    scenes -- at high contrast, plus a dense layer of small high-contrast texture elements (speckles, short strokes, checker patches: foliage /
    gravel / brick-like detail).  Several times the FAST candidates of make_canvas per frame, corners crowded next to each other (NMS ties,
    full candidate queues), few flat cells.  The stress scene of the constants the kernels were tuned on the quad scene with (queue
    capacities, quad-tree tier thresholds, the share of cells that take the minThFAST pass)."""
    rng = np.random.default_rng(seed)
    fy = np.fft.fftfreq(size)[:, None]
    fx = np.fft.rfftfreq(size)[None, :]
    f = np.sqrt(fx * fx + fy * fy)
    f[0, 0] = 1.0
    spec = (rng.normal(size=f.shape) + 1j * rng.normal(size=f.shape)) / f       # amplitude ~ 1 / f
    spec[0, 0] = 0.0
    pink = np.fft.irfft2(spec, s=(size, size)).astype(np.float32)
    pink = (pink - pink.mean()) / (pink.std() + 1e-9)
    canvas = np.clip(128.0 + 46.0 * pink, 0, 255).astype(np.float32)
    # dense texture elements: ~1 per 12 x 12 px
    n_el = (size // 12) ** 2
    cx = rng.integers(2, size - 10, n_el)
    cy = rng.integers(2, size - 10, n_el)
    kind = rng.integers(0, 3, n_el)
    grey = np.where(rng.random(n_el) < 0.5, rng.uniform(0, 70, n_el), rng.uniform(185, 255, n_el)).astype(np.float32)
    ew = rng.integers(1, 6, n_el)
    eh = rng.integers(1, 6, n_el)
    for i in range(n_el):
        x, y = int(cx[i]), int(cy[i])
        if kind[i] == 0:      # speckle / blob
            canvas[y:y + eh[i], x:x + ew[i]] = grey[i]
        elif kind[i] == 1:    # short stroke
            if ew[i] >= eh[i]:
                canvas[y:y + 1 + (eh[i] > 3), x:x + 2 * ew[i]] = grey[i]
            else:
                canvas[y:y + 2 * eh[i], x:x + 1 + (ew[i] > 3)] = grey[i]
        else:                 # 2 x 2 checker of cells ew x eh
            canvas[y:y + eh[i], x:x + ew[i]] = grey[i]
            canvas[y + eh[i]:y + 2 * eh[i], x + ew[i]:x + 2 * ew[i]] = grey[i]
            canvas[y:y + eh[i], x + ew[i]:x + 2 * ew[i]] = 255.0 - grey[i]
            canvas[y + eh[i]:y + 2 * eh[i], x:x + ew[i]] = 255.0 - grey[i]
    return canvas


def frame_from_canvas(canvas: np.ndarray, t: int, w: int, h: int, noise_seed: int, sigma: float = 3.0,
                      dx: int = 2, dy: int = 1, x0: int = 0, y0: int = 0) -> np.ndarray:
    size = canvas.shape[0]
    ox = (x0 + dx * t) % (size - w)
    oy = (y0 + dy * t) % (size - h)
    crop = canvas[oy:oy + h, ox:ox + w]
    rng = np.random.default_rng(noise_seed)
    noisy = crop + rng.normal(0.0, sigma, crop.shape).astype(np.float32)
    return np.clip(np.rint(noisy), 0, 255).astype(np.uint8)


def make_frames(seed: int, n: int, w: int, h: int, canvas: np.ndarray | None = None) -> np.ndarray:
    """(n, h, w) uint8 frames of scene `seed` (frame t = crop at (2t, t) + noise seeded 1000*seed+t)."""
    if canvas is None:
        canvas = make_canvas(seed, size=max(2048, 2 * max(w, h)))
    return np.stack([frame_from_canvas(canvas, t, w, h, 1000 * seed + t) for t in range(n)])


def make_stereo_pair(seed: int, t: int, w: int, h: int, canvas: np.ndarray | None = None):
    """Rectified stereo pair: the right image sees the canvas shifted by a per-band disparity."""
    if canvas is None:
        canvas = make_canvas(seed, size=max(2048, 2 * max(w, h)))
    left = frame_from_canvas(canvas, t, w, h, 3000 + t, x0=128)
    right = np.empty_like(left)
    bands = [8, 16, 32, 64]
    bh = h // len(bands)
    for k, d in enumerate(bands):
        ys = slice(k * bh, h if k == len(bands) - 1 else (k + 1) * bh)
        full = frame_from_canvas(canvas, t, w, h, 4000 + t, x0=128 + d)
        right[ys] = full[ys]
    return left, right


def make_test_image(seed: int, w: int, h: int) -> np.ndarray:
    """Small structured + noisy image for unit tests (fast to build)."""
    canvas = make_canvas(seed, size=max(512, 2 * max(w, h)), n_shapes=150)
    return frame_from_canvas(canvas, 0, w, h, seed + 77)


def make_scene_canvas(scene: str, seed: int, size: int = 2048) -> np.ndarray:
    """Canvas by scene name: "quads" (make_canvas: SURVEY.md 8(d)'s scene, the metric), "texture" (make_texture_canvas: 13 x the FAST candidates), or
    "blend:<a>" with 0 <= a <= 1: (1 - a) * quads + a * texture, pixel by pixel -- candidate densities in between (bench.py --scene, tools/density_sweep.sh)."""
    if scene == "quads":
        return make_canvas(seed, size=size) if size != 2048 else make_canvas(seed)
    if scene == "texture":
        return make_texture_canvas(seed, size)
    if scene.startswith("blend:"):
        a = float(scene.split(":", 1)[1])
        if not 0.0 <= a <= 1.0:
            raise ValueError("blend factor must lie in [0, 1]")
        q = make_canvas(seed, size=size) if size != 2048 else make_canvas(seed)
        t = make_texture_canvas(seed, size)
        return ((1.0 - a) * q.astype(np.float32) + a * t.astype(np.float32)).astype(np.float32)
    raise ValueError(f"unknown scene {scene!r}")


# Examples/Stereo/TUM-VI.yaml Camera1 / Camera2: fx, fy, cx, cy, k0..k3 (KannalaBrandt8::mvParameters)
TUMVI_L = (190.978477, 190.973307, 254.931706, 256.897442, 0.0034823894, 0.0007150348, -0.0020532361, 0.0002029367)
TUMVI_R = (190.442369, 190.434438, 252.598164, 254.917230, 0.0034003171, 0.0017669271, -0.0026631290, 0.0003299517)


def make_fisheye_keyframes(rng, n_pts: int = 420, with_poses: bool = False):
    """Two key frames of a TUM-VI-like fisheye rig looking at common 3-D points: features = mvKeys | mvKeysRight, descriptors noisy copies of the point's,
    and the four relative poses of ORBmatcher.cc:934-944 (ll, lr, rl, rr).  Returns (k1, n_left1, d1, id1, k2, n_left2, d2, id2, R12, t12, cams)."""
    from . import KP_DTYPE

    def rot(a):
        ax, ay, az = a
        Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
        Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
        Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
        return Rz @ Ry @ Rx

    def se3(R, t):
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
        return T
    Trl = se3(rot(rng.uniform(-0.01, 0.01, 3)), [-0.101, 0.002, 0.001])     # right camera from left camera
    Tw2 = se3(rot(rng.uniform(-0.04, 0.04, 3)), [0.25, 0.03, -0.02])         # KF2's left camera in the world = KF1's left camera frame (T1w = I)
    Tlr = np.linalg.inv(Trl)
    pair = [Tw2, Tw2 @ Tlr, Trl @ Tw2, Trl @ Tw2 @ Tlr]                      # Tll, Tlr, Trl, Trr: x_cam1 = T * x_cam2
    R12 = np.stack([T[:3, :3] for T in pair]).astype(np.float32)
    t12 = np.stack([T[:3, 3] for T in pair]).astype(np.float32)
    P = np.stack([rng.uniform(-3, 3, n_pts), rng.uniform(-3, 3, n_pts), rng.uniform(1.0, 8, n_pts), np.ones(n_pts)], axis=1)
    pdesc = rng.integers(0, 256, (n_pts, 32), dtype=np.uint8)

    def proj(prm, X):
        th = np.arctan2(np.hypot(X[:, 0], X[:, 1]), X[:, 2]); psi = np.arctan2(X[:, 1], X[:, 0])
        r = th + prm[4] * th ** 3 + prm[5] * th ** 5 + prm[6] * th ** 7 + prm[7] * th ** 9
        return np.stack([prm[0] * r * np.cos(psi) + prm[2], prm[1] * r * np.sin(psi) + prm[3]], axis=1)

    def keyframe(Tcw_left):
        parts, ids = [], []
        for T, prm in ((Tcw_left, TUMVI_L), (Trl @ Tcw_left, TUMVI_R)):
            X = (T @ P.T).T[:, :3]
            uv = proj(prm, X) + rng.choice([0.2, 1.0, 5.0], n_pts)[:, None] * rng.uniform(-1, 1, (n_pts, 2))
            seen = np.nonzero((rng.random(n_pts) < 0.7) & (uv[:, 0] > 5) & (uv[:, 0] < 507) & (uv[:, 1] > 5) & (uv[:, 1] < 507))[0]
            k = np.zeros(len(seen), KP_DTYPE)
            k["x"], k["y"] = uv[seen, 0], uv[seen, 1]
            k["octave"] = rng.integers(0, 8, len(seen))
            k["angle"] = rng.uniform(0, 360, len(seen))
            k["size"], k["class_id"] = 31.0, -1
            parts.append(k); ids.append(seen)
        kps = np.concatenate(parts)
        pid = np.concatenate(ids)
        flip = rng.random((len(pid), 256)) < 0.04
        return kps, len(parts[0]), pdesc[pid] ^ np.packbits(flip, axis=1, bitorder="little"), pid
    k1, nl1, d1, id1 = keyframe(np.eye(4))
    k2, nl2, d2, id2 = keyframe(np.linalg.inv(Tw2))
    out = (k1, nl1, d1, id1, k2, nl2, d2, id2, R12, t12, np.array([TUMVI_L, TUMVI_R], np.float32))
    if with_poses:   # Tcw of the two left cameras and Trl, each (R, t) in float32: what a KeyFrame holds (GetPose, GetRelativePoseTrl)
        T2w = np.linalg.inv(Tw2)
        f = lambda T: (T[:3, :3].astype(np.float32), T[:3, 3].astype(np.float32))
        out += (dict(pose1=f(np.eye(4)), pose2=f(T2w), trl=f(Trl)),)
    return out
