"""Seeded synthetic frame generator.

No dataset is available offline, so every config of BASELINE.json runs on synthetic
camera-like frames: a large canvas (smooth background + random filled convex quads of
random grey levels, i.e. plenty of FAST corners at every pyramid scale) from which frame t
is the crop at offset (2t, t) plus per-frame Gaussian sensor noise.  Pure numpy; used by
tests, bench.py and __graft_entry__.smoke().
"""
from __future__ import annotations

import numpy as np

__all__ = ["make_canvas", "frame_from_canvas", "make_frames", "make_stereo_pair", "make_test_image"]


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
