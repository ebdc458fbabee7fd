"""Real-dataset input for the bench workloads: when ORBX_EUROC_DIR / ORBX_KITTI_DIR /
ORBX_TUMVI_DIR point at a sequence folder, its first frames replace the synthetic scene.

  EuRoC   <dir>/mav0/cam0/data/*.png            (also accepted: <dir>/cam0/data, <dir> itself)      752x480 8-bit gray
  KITTI   <dir>/image_0/*.png + <dir>/image_1/*.png  (odometry sequence folder, rectified gray pair)  1241x376
  TUM-VI  <dir>/mav0/cam0/data/*.png            (16-bit PNGs are reduced to their high byte)         1024x1024 ("_1024" sequences)

Images are centre-cropped / rejected so that every frame has the workload's shape.  PNG decoding: Pillow (present in the image).
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import List, Optional

import numpy as np

ENV = {"euroc": "ORBX_EUROC_DIR", "kitti": "ORBX_KITTI_DIR", "tumvi": "ORBX_TUMVI_DIR"}


def dataset_dir(kind: str) -> Optional[Path]:
    v = os.environ.get(ENV[kind])
    return Path(v) if v else None


def read_gray(path: Path) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        if im.mode in ("I;16", "I;16B", "I"):
            a = np.asarray(im)
            return (a.astype(np.uint32) >> 8).astype(np.uint8) if a.max() > 255 else a.astype(np.uint8)
        return np.asarray(im.convert("L"), np.uint8)


def _image_folder(root: Path, candidates: List[str]) -> Path:
    for c in candidates:
        d = root / c
        if d.is_dir() and any(d.glob("*.png")):
            return d
    raise FileNotFoundError(f"no PNG folder under {root} (looked for {candidates})")


def _fit(img: np.ndarray, w: int, h: int) -> np.ndarray:
    H, W = img.shape
    if W < w or H < h:
        raise ValueError(f"frame {W}x{H} smaller than the workload's {w}x{h}")
    y0, x0 = (H - h) // 2, (W - w) // 2
    return np.ascontiguousarray(img[y0:y0 + h, x0:x0 + w])


def load_mono(kind: str, n: int, w: int, h: int, start: int = 0) -> np.ndarray:
    """(n, h, w) uint8: frames start .. start+n-1 of the sequence (cycled if the folder holds fewer)."""
    root = dataset_dir(kind)
    folder = _image_folder(root, ["mav0/cam0/data", "cam0/data", "image_0", "."])
    files = sorted(folder.glob("*.png"))
    return np.stack([_fit(read_gray(files[(start + t) % len(files)]), w, h) for t in range(n)])


def load_stereo(kind: str, n: int, w: int, h: int):
    """[(left, right)] * n for a rectified stereo sequence (KITTI odometry layout, or EuRoC cam0 / cam1)."""
    root = dataset_dir(kind)
    left = _image_folder(root, ["image_0", "mav0/cam0/data"])
    right = _image_folder(root, ["image_1", "mav0/cam1/data"])
    fl, fr = sorted(left.glob("*.png")), sorted(right.glob("*.png"))
    m = min(len(fl), len(fr))
    return [(_fit(read_gray(fl[t % m]), w, h), _fit(read_gray(fr[t % m]), w, h)) for t in range(n)]
