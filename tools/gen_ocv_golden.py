#!/usr/bin/env python3
"""Pins the oracle's restated OpenCV primitives against a REAL OpenCV, on any machine that has one.

The reference's path runs six OpenCV primitives (cv::resize INTER_LINEAR 8U, copyMakeBorder REFLECT_101, FAST 9/16 + NMS,
GaussianBlur 7x7 sigma 2, fastAtan2, undistortPoints) whose arithmetic the oracle restates from the published algorithms because
OpenCV is absent from the reference tree and from the build image ("parity unpinned").  This script needs
`cv2` (any OpenCV >= 4.4): it runs the real primitives on seeded inputs and writes tests/golden/ocv_primitives.npz (inputs are
regenerated from the seeds; only outputs are stored).  tests/test_oracle_vs_opencv.py compares the oracle with that file when it
exists -- or with cv2 directly when it is importable -- and skips with "parity unpinned" otherwise.

    python tools/gen_ocv_golden.py          # on a box with opencv-python; then commit tests/golden/ocv_primitives.npz
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
OUT = ROOT / "tests" / "golden" / "ocv_primitives.npz"


def inputs():
    """Seeded inputs shared by the generator and the test (pure numpy, no OpenCV)."""
    from orb_slam3_amd import synth
    img = synth.frame_from_canvas(synth.make_canvas(1), 3, 752, 480, 1003)
    small = synth.make_test_image(5, 320, 240)
    rng = np.random.default_rng(7)
    yx = rng.normal(0, 300, (4096, 2)).astype(np.float32)
    yx[:8] = [[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1], [1, 1], [-1, 1], [1e-20, 1]]
    pts = np.stack([rng.uniform(0, 752, 2000), rng.uniform(0, 480, 2000)], axis=1).astype(np.float32)
    return dict(img=img, small=small, yx=yx, pts=pts)


EUROC_K = (458.654, 457.296, 367.215, 248.375)
EUROC_D = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)


def run_opencv(cv2, inp):
    out = {"version": np.array([int(x) for x in cv2.__version__.split(".")[:3]])}
    img, small = inp["img"], inp["small"]
    lvl = img
    for l in range(1, 8):   # the pyramid chain of ORBextractor::ComputePyramid (:1171-1195): resize from the previous level
        w, h = int(round(752 / 1.2 ** l)), int(round(480 / 1.2 ** l))
        lvl = cv2.resize(lvl, (w, h), interpolation=cv2.INTER_LINEAR)
        out[f"resize{l}"] = lvl
    out["border"] = cv2.copyMakeBorder(small, 19, 19, 19, 19, cv2.BORDER_REFLECT_101)
    out["blur"] = cv2.GaussianBlur(img, (7, 7), 2, sigmaY=2, borderType=cv2.BORDER_REFLECT_101)
    for th in (20, 7):
        det = cv2.FastFeatureDetector_create(threshold=th, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
        kps = det.detect(small[16:80, 16:90].copy(), None)   # a cell-sized sub-image, as ComputeKeyPointsOctTree calls it
        kps2 = det.detect(small, None)
        for name, k in ((f"fast_cell{th}", kps), (f"fast_full{th}", kps2)):
            out[name] = np.array([[p.pt[0], p.pt[1], p.response] for p in k], np.float32).reshape(-1, 3)
    out["atan2"] = np.array([cv2.fastAtan2(float(y), float(x)) for y, x in inp["yx"]], np.float32)
    K = np.array([[EUROC_K[0], 0, EUROC_K[2]], [0, EUROC_K[1], EUROC_K[3]], [0, 0, 1]], np.float64)
    D = np.array(EUROC_D, np.float64)
    out["undistort"] = cv2.undistortPoints(inp["pts"].reshape(-1, 1, 2), K, D, R=None, P=K).reshape(-1, 2).astype(np.float32)
    return out


def main():
    try:
        import cv2
    except ImportError:
        sys.exit("gen_ocv_golden.py needs OpenCV's Python module (cv2); it is not installed here: parity of the OpenCV primitives stays unpinned")
    out = run_opencv(cv2, inputs())
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} from OpenCV {cv2.__version__}")


if __name__ == "__main__":
    main()
