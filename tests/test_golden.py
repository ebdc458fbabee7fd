"""CPU tests: the oracle against the committed golden fixtures (tests/golden/*.npz, made by tools/gen_golden.py
from the oracle at the commit that introduced them -- regression pins, since the reference holds no vectors)."""
from pathlib import Path

import numpy as np

GOLD = Path(__file__).resolve().parent / "golden"


def test_extractor_golden(oracle):
    from orb_slam3_amd import synth
    g = np.load(GOLD / "extract_320x240_seed5.npz")
    img = synth.make_test_image(5, 320, 240)
    assert np.array_equal(img, g["image"])
    for flags, tag in ((oracle.FLAG_DESC_FMA, "fma"), (0, "strict")):
        ex = oracle.OracleExtractor(500, 1.2, 8, 20, 7, flags=flags)
        mono, kps, desc = ex.extract(img, lap=(0, 1000))
        assert mono == int(g["mono"])
        assert kps.tobytes() == g["kps"].tobytes()
        assert np.array_equal(desc, g["desc_" + tag])
    assert np.array_equal(ex.level_padded(3), g["level3"])
    assert np.array_equal(ex.level_blurred(2), g["blur2"])


def test_matcher_golden(oracle):
    g = np.load(GOLD / "match_seed7.npz")
    idx, dist = oracle.knn2(g["q"], g["t"])
    assert np.array_equal(idx, g["knn_idx"]) and np.array_equal(dist, g["knn_dist"])
