"""Edge-case sweep of the matcher oracle against the reference's own ORBmatcher.cc (oracle/_ref/libmatcher_ref.so): hundreds of tiny
random problems per matcher -- empty frames, a single feature, every slot occupied, queries on and beyond the image border, windows
larger than the image, top / bottom pyramid levels, duplicate descriptors, empty vocabulary nodes.  Live when the compiled reference
is here, else against its committed outputs (tests/golden/matchers_small_ref.npz)."""
import numpy as np
import pytest

from oracle import oracle_binding as ob
from oracle import ref_binding as rb
from orb_slam3_amd.matcher import FeatureVector
from _pin import Pinner

_P = Pinner("matchers_small_ref.npz", rb.matcher_available())
W, H = 160.0, 120.0
SF = np.array([1.2 ** i for i in range(8)], np.float32)
ISG = (1.0 / (SF.astype(np.float64) ** 2)).astype(np.float32)


@pytest.fixture(scope="module", autouse=True)
def _write_golden():
    yield
    _P.finish()


def _kps(rng, n, spread=0.0):
    k = np.zeros(n, ob.KP_DTYPE)
    k["x"] = rng.uniform(-spread, W + spread, n)
    k["y"] = rng.uniform(-spread, H + spread, n)
    k["octave"] = rng.integers(0, 8, n)
    k["angle"] = rng.uniform(0, 360, n)
    return k


def _desc(rng, n, pool):
    """descriptors drawn from a small pool with a few flipped bits: many near and exact duplicates"""
    if n == 0:
        return np.zeros((0, 32), np.uint8)
    d = pool[rng.integers(0, len(pool), n)].copy()
    flips = rng.random((n, 256)) < rng.choice([0.0, 0.02, 0.1])
    return d ^ np.packbits(flips, axis=1, bitorder="little")


def _sizes(rng):
    return int(rng.choice([0, 1, 2, 5, 17, 40])), int(rng.choice([0, 1, 3, 9, 33, 70]))


def test_projection_matchers_small_cases():
    rng = np.random.default_rng(100)
    pool = rng.integers(0, 256, (6, 32), dtype=np.uint8)
    outs_o, calls = [], []
    for case in range(250):
        n, nq = _sizes(rng)
        k, d = _kps(rng, n, spread=6.0), _desc(rng, n, pool)
        grid = ob.OracleGrid(k, 0.0, W, 0.0, H)
        F = rb.RefFrame(k, d, 0.0, W, 0.0, H, SF, None)
        occ = (rng.random(n) < rng.choice([0.0, 0.3, 1.0])).astype(np.uint8)
        qx = rng.uniform(0, W, nq).astype(np.float32)
        qy = rng.uniform(0, H, nq).astype(np.float32)
        if nq and n and rng.random() < 0.5:       # queries sitting exactly on features, on the border, in the corner
            j = rng.integers(0, n, nq)
            qx, qy = np.clip(k["x"][j], 0, W - 1e-3).astype(np.float32), np.clip(k["y"][j], 0, H - 1e-3).astype(np.float32)
        if nq:
            qx[0], qy[0] = 0.0, 0.0
        lvl = rng.integers(0, 8, nq).astype(np.int32)
        dq = _desc(rng, nq, pool)
        ang = rng.uniform(0, 360, nq).astype(np.float32)
        th = float(rng.choice([1.0, 3.0, 15.0, 400.0]))
        # M1
        mp = dict(proj_x=qx, proj_y=qy, proj_xr=qx - 3.0, level=lvl, view_cos=rng.choice([0.99, 0.9995], nq).astype(np.float32), desc=dq,
                  in_view=(rng.random(nq) < 0.9).astype(np.uint8), has_obs=(rng.random(nq) < 0.8).astype(np.uint8))
        ratio = float(rng.choice([0.6, 0.8, 0.9]))
        outs_o.append(ob.search_by_projection_mappoints(grid, d, SF, mp, th, ratio, None, occ))
        calls.append(lambda F=F, mp=mp, th=th, ratio=ratio, occ=occ: rb.ref_search_by_projection_mappoints(F, mp, th, ratio, occ))
        # M2 in its three level modes
        z = (rng.integers(8, 320, nq) / 8.0).astype(np.float32)
        q2 = dict(u=qx, v=qy, z=z, ur=(qx - np.float32(1.0) / z).astype(np.float32), octave=lvl, angle=ang, desc=dq,
                  has_obs=(rng.random(nq) < 0.8).astype(np.uint8))
        mode, ori = int(rng.integers(0, 3)), bool(rng.integers(0, 2))
        outs_o.append(ob.search_by_projection_frame(grid, d, SF, q2, th, mode, ori, None, occ))
        calls.append(lambda F=F, q2=q2, th=th, mode=mode, ori=ori, occ=occ: rb.ref_search_by_projection_frame(F, q2, th, mode, ori, occ))
        # M3 (Frame grid, [l-1, l+1]) and M4 (KeyFrame grid + octave gate [l-1, l])
        r = (np.float32(th) * SF[lvl]).astype(np.float32)
        q3 = dict(x=qx, y=qy, r=r, min_level=lvl - 1, max_level=lvl + 1, angle=ang, desc=dq)
        orbdist = int(rng.choice([64, 100]))
        outs_o.append(ob.search_by_projection_window(grid, d, q3, float(orbdist), ori, False, occ))
        calls.append(lambda F=F, qx=qx, qy=qy, lvl=lvl, ang=ang, dq=dq, th=th, orbdist=orbdist, ori=ori, occ=occ:
                     rb.ref_search_by_projection_keyframe(F, dict(x=qx, y=qy, level=lvl, angle=ang, desc=dq), th, orbdist, ori, occ))
        thi = int(rng.choice([3, 5, 8]))
        rh = float(rng.choice([1.0, 1.5]))
        q4 = dict(x=qx, y=qy, r=(np.float32(thi) * SF[lvl]).astype(np.float32), min_level=lvl - 1, max_level=lvl, desc=dq)
        outs_o.append(ob.search_by_projection_window(grid, d, q4, 50 * rh, False, True, occ))
        calls.append(lambda F=F, qx=qx, qy=qy, lvl=lvl, dq=dq, thi=thi, rh=rh, occ=occ, v=case & 1:
                     rb.ref_search_by_projection_sim3(F, dict(x=qx, y=qy, level=lvl, desc=dq), thi, rh, v, occ))
        # Fuse x2
        qf = dict(u=qx, v=qy, z=z, ur=(qx - np.float32(1.0) / z).astype(np.float32), r=r, level=lvl, desc=dq)
        ur = np.where(rng.random(n) < 0.5, k["x"] - 2.0, -1.0).astype(np.float32) if rng.random() < 0.5 else None
        Fu = rb.RefFrame(k, d, 0.0, W, 0.0, H, SF, ur)
        for variant in (0, 1):
            bi, bd = ob.fuse_search(grid, d, ur if variant == 0 else None, ISG if variant == 0 else None, qf, fma=True)
            want = np.where(bd <= 50, bi, -1)
            outs_o.append(((want >= 0).sum(), want))
            calls.append(lambda Fu=Fu, qf=qf, th=th, variant=variant: rb.ref_fuse(Fu, ISG, qf, th, variant))
    _P.pin("projection", [x for o in outs_o for x in o], lambda: [x for c in calls for x in c()])


def test_bow_and_initialization_small_cases():
    rng = np.random.default_rng(200)
    pool = rng.integers(0, 256, (5, 32), dtype=np.uint8)
    outs_o, calls = [], []
    for case in range(250):
        n1, n2 = _sizes(rng)
        k1, k2 = _kps(rng, n1), _kps(rng, n2)
        d1, d2 = _desc(rng, n1, pool), _desc(rng, n2, pool)
        nn = int(rng.choice([1, 3, 10]))
        fv1 = FeatureVector.from_node_of_feature(rng.integers(0, nn, n1) * 2)      # even ids on one side, all ids on the other:
        fv2 = FeatureVector.from_node_of_feature(rng.integers(0, 2 * nn, n2))      # nodes present on one side only
        v1 = (rng.random(n1) < rng.choice([0.5, 1.0])).astype(np.uint8)
        v2 = (rng.random(n2) < rng.choice([0.5, 1.0])).astype(np.uint8)
        ratio, ori = float(rng.choice([0.6, 0.75, 0.9])), bool(rng.integers(0, 2))
        a1, a2 = k1["angle"], k2["angle"]
        outs_o.append(ob.search_by_bow_frame(d1, a1, v1, fv1, d2, a2, fv2, ratio, ori))
        calls.append(lambda a=(d1, a1, v1, fv1, d2, a2, fv2, ratio, ori): rb.ref_search_by_bow_frame(*a))
        outs_o.append(ob.search_by_bow_keyframes(d1, a1, v1, fv1, d2, a2, v2, fv2, ratio, ori))
        calls.append(lambda a=(d1, a1, v1, fv1, d2, a2, v2, fv2, ratio, ori): rb.ref_search_by_bow_keyframes(*a))
        s1, s2 = 1 - v1, 1 - v2
        tab = (rng.random((n1, n2)) < 0.7).astype(np.uint8)
        coarse = bool(rng.integers(0, 2))
        outs_o.append(ob.search_for_triangulation(d1, a1, s1, fv1, d2, a2, s2, fv2, ori, None if coarse else (lambda i, j, tab=tab: tab[i, j])))
        calls.append(lambda a=(d1, a1, s1, fv1, d2, a2, s2, fv2, ori, tab, coarse): rb.ref_search_for_triangulation(*a))
        # M6: level-0 keypoints are the only queries
        k1["octave"] = rng.integers(0, 2, n1)
        k2["octave"] = rng.integers(0, 2, n2)
        win = int(rng.choice([5, 30, 100]))
        prev_a = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)).reshape(n1, 2)
        prev_b = prev_a.copy()
        grid2 = ob.OracleGrid(k2, 0.0, W, 0.0, H)
        on, om = ob.search_for_initialization(k1, d1, grid2, d2, prev_a, win, ratio, ori)
        outs_o.append((on, om, prev_a))
        calls.append(lambda a=(k1.copy(), d1, k2.copy(), d2, np.array([0, W, 0, H], np.float32), prev_b, win, ratio, ori):
                     rb.ref_search_for_initialization(*a) + (a[5],))
    _P.pin("bow_init", [x for o in outs_o for x in o], lambda: [x for c in calls for x in c()])
