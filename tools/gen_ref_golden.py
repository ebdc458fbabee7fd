#!/usr/bin/env python3
"""Golden outputs of the REFERENCE (oracle/_ref/liborb_ref.so = /root/reference/src/ORBextractor.cc compiled against
oracle/ocv_shim) for tests/test_oracle_vs_reference.py::test_oracle_equals_reference_golden.  Run where /root/reference exists:
    make -C oracle ref && python tools/gen_ref_golden.py"""
import hashlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import ref_binding as rb  # noqa: E402
from orb_slam3_amd import synth  # noqa: E402

CASES = {  # small enough to commit: keypoints 28 B + descriptors 32 B each
    "ref_euroc_752x480": ((752, 480, 1000, 1.2, 8, 20, 7, 1), (0, 1000)),
    "ref_small_320x240": ((320, 240, 300, 1.2, 6, 20, 7, 8), (0, 0)),
    "ref_lapping_640x400": ((640, 400, 500, 1.2, 8, 20, 7, 9), (150, 420)),
}
for name, (case, lap) in CASES.items():
    w, h, nf, sf, nl, ini, mn, seed = case
    img = synth.make_test_image(seed, w, h)
    mono, k, d = rb.RefExtractor(nf, sf, nl, ini, mn).extract(img, lap)
    np.savez_compressed(ROOT / "tests" / "golden" / f"{name}.npz", case=np.array(case, np.float64), lap=np.array(lap), mono=mono,
                        keypoints=k.view(np.uint8).reshape(len(k), 28), descriptors=d,
                        image_sha256=hashlib.sha256(img.tobytes()).hexdigest())
    print(name, mono, len(k))
