"""The oracle against the REFERENCE ITSELF: /root/reference/src/ORBextractor.cc compiled where it lies (oracle/Makefile,
`make ref`) against oracle/ocv_shim, a stand-in for the OpenCV API whose five arithmetic primitives forward to the oracle's
restated ones.  This pins every line of the reference's own code path (tables, pyramid composition, per-cell FAST with the
threshold fallback, quad-tree, orientation, steered BRIEF incl. the FMA contraction of the reference's build flags, output
assembly and lapping split) bit for bit; the OpenCV primitives themselves stay [OCV-recalled].

Two layers: live comparison when oracle/_ref/*.so is present (built in the build container, travels with gpurun), and
committed golden outputs of that library (tests/golden/ref_*.npz, made by tools/gen_ref_golden.py) that need nothing but
the oracle."""
import hashlib
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle_binding as ob
from oracle import ref_binding as rb
from orb_slam3_amd import synth

GOLDEN = Path(__file__).resolve().parent / "golden"

CASES = [  # (w, h, nfeatures, scaleFactor, nlevels, iniTh, minTh, seed)
    (752, 480, 1000, 1.2, 8, 20, 7, 1),      # EuRoC
    (1241, 376, 2000, 1.2, 8, 20, 7, 2),     # KITTI
    (512, 512, 1500, 1.2, 8, 20, 7, 3),      # TUM-VI (shipped YAML size)
    (752, 480, 5000, 1.2, 8, 20, 7, 4),      # monocular initialisation extractor (5 x nFeatures)
    (641, 479, 300, 1.5, 5, 40, 12, 5),      # odd sizes, coarse pyramid, high thresholds
    (800, 600, 1500, 1.1, 10, 15, 5, 6),     # fine pyramid
    (1600, 300, 1200, 1.2, 4, 20, 7, 7),     # panorama: six quad-tree roots
]
LAPS = [(0, 0), (0, 1000), (200, 400)]


def _image(w, h, seed):
    return synth.make_test_image(seed, w, h)


needs_ref = pytest.mark.skipif(not rb.available(), reason="oracle/_ref not built (needs /root/reference at build time)")


@needs_ref
@pytest.mark.parametrize("case", CASES)
def test_oracle_equals_compiled_reference(case):
    w, h, nf, sf, nl, ini, mn, seed = case
    img = _image(w, h, seed)
    ref = rb.RefExtractor(nf, sf, nl, ini, mn)
    orc = ob.OracleExtractor(nf, sf, nl, ini, mn, flags=ob.FLAG_DESC_FMA)
    rt, ot = ref.tables(), orc.tables()
    for k in ("scale", "inv_scale", "sigma2", "inv_sigma2"):
        assert np.array_equal(rt[k], ot[k]), k
    for lap in LAPS:
        rm, rk, rd = ref.extract(img, lap)
        om, ok, od = orc.extract(img, lap=lap)
        assert rm == om and len(rk) == len(ok) and len(rk) > 50
        assert np.array_equal(rk, ok) and np.array_equal(rd, od), (case, lap)
    for level in range(nl):
        assert np.array_equal(ref.level_padded(level), orc.level_padded(level)), level   # mvImagePyramid incl. the 19-px ring


@needs_ref
def test_strict_float_build_of_the_reference():
    """-ffp-contract=off build of the same reference source = the oracle's strict descriptor mode (ORBX_FLAG_DESC_STRICT)."""
    w, h, nf, sf, nl, ini, mn, seed = CASES[0]
    img = _image(w, h, seed)
    rm, rk, rd = rb.RefExtractor(nf, sf, nl, ini, mn, strict=True).extract(img, (0, 1000))
    om, ok, od = ob.OracleExtractor(nf, sf, nl, ini, mn, flags=0).extract(img, lap=(0, 1000))
    assert rm == om and np.array_equal(rk, ok) and np.array_equal(rd, od)
    fm, fk, fd = ob.OracleExtractor(nf, sf, nl, ini, mn, flags=ob.FLAG_DESC_FMA).extract(img, lap=(0, 1000))
    assert np.array_equal(fk, ok)   # the build flags can only change descriptor bits (rarely: a sample coordinate at x.5)


@pytest.mark.parametrize("name", sorted(p.name for p in GOLDEN.glob("ref_*.npz")))
def test_oracle_equals_reference_golden(name):
    """Outputs of the compiled reference stored in the repository (no /root/reference, no _ref needed)."""
    g = np.load(GOLDEN / name)
    w, h, nf, sf, nl, ini, mn, seed = [g["case"][i] for i in range(8)]
    img = _image(int(w), int(h), int(seed))
    assert hashlib.sha256(img.tobytes()).hexdigest() == str(g["image_sha256"])
    orc = ob.OracleExtractor(int(nf), float(sf), int(nl), int(ini), int(mn), flags=ob.FLAG_DESC_FMA)
    mono, k, d = orc.extract(img, lap=(int(g["lap"][0]), int(g["lap"][1])))
    assert mono == int(g["mono"])
    assert np.array_equal(k, g["keypoints"].view(ob.KP_DTYPE).reshape(-1)) and np.array_equal(d, g["descriptors"])


def _ordered_vocabulary(rng, k, L):
    """Random k-ary vocabulary (2..k children per node, all leaves at depth L as in ORBvoc.txt -- for a leaf above the level
    asked for the reference leaves *nid unassigned) numbered the way TemplatedVocabulary::loadFromTextFile numbers it: a
    node's id is its line number, so children lists are ascending; word ids follow leaf id order.  Sibling descriptors are
    duplicated in places so that distance ties occur (the first child must win)."""
    children = [[]]
    depth = [0]
    q = [0]
    while q:
        i = q.pop(0)
        if depth[i] == L:
            continue
        for _ in range(int(rng.integers(max(2, k - 3), k + 1))):
            children.append([]); depth.append(depth[i] + 1)
            children[i].append(len(children) - 1)
            q.append(len(children) - 1)
    n = len(children)
    nd = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for ch in children:
        if len(ch) >= 2 and rng.random() < 0.15:
            nd[ch[1]] = nd[ch[0]]
    cp, ci, wi, nw = [0], [], np.full(n, -1, np.int32), 0
    for i in range(n):
        ci.extend(children[i]); cp.append(len(ci))
        if not children[i]:
            wi[i] = nw; nw += 1
    return np.array(cp, np.int32), np.array(ci, np.int32), nd, wi


@pytest.mark.skipif(not rb.dbow_available(), reason="oracle/_ref/libdbow2_ref.so not built (needs /root/reference at build time)")
def test_bow_transform_equals_vendored_dbow2(tmp_path):
    """orbo_bow_transform (and through it orbx_bow_transform, tests/test_gpu_matcher.py::test_bow_transform) against the
    reference's vendored DBoW2: TemplatedVocabulary<FORB::TDescriptor, FORB>::transform on a vocabulary loaded with
    loadFromTextFile, per feature (word, node at levelsup) and the FeatureVector of the batch overload Frame::ComputeBoW calls."""
    rng = np.random.default_rng(77)
    for k, L in ((10, 4), (6, 6), (17, 3)):
        cp, ci, nd, wi = _ordered_vocabulary(rng, k, L)
        path = tmp_path / f"voc_{k}_{L}.txt"
        rb.write_vocabulary_text(path, k, L, cp, ci, nd, wi)
        voc = rb.RefVocabulary(path)
        assert voc.words() == int((wi >= 0).sum())
        feats = np.concatenate([rng.integers(0, 256, (700, 32), dtype=np.uint8), nd[rng.integers(1, len(nd), 300)]])
        for levelsup in (4, 2, 0, 9):
            rw, rn, fvn, fvf = voc.transform(feats, levelsup)
            ow, on = ob.bow_transform(cp, ci, nd, wi, L, levelsup, feats)
            assert np.array_equal(rw, ow) and np.array_equal(rn, on), (k, L, levelsup)
            # FeatureVector = features grouped by node id (ascending), feature order kept inside a node
            order = np.lexsort((np.arange(len(on)), on))
            assert np.array_equal(fvn, on[order]) and np.array_equal(fvf, order)
