"""ctypes binding of the CPU oracle (oracle/liborb_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (orb_slam3_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_SO = _DIR / "liborb_oracle.so"

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

FLAG_DESC_FMA = 1
FLAG_BLUR_OCV440 = 2
FLAG_LIBM_SINCOS = 4
FLAG_ATAN_FMA = 8


def build(force: bool = False) -> Path:
    """Compile the oracle with g++ (seconds).  The GPU box uses the prebuilt .so if present."""
    srcs = [_DIR / "orb_oracle.cc", _DIR / "orb_oracle_match.cc", _DIR / "orb_oracle_geom.cc", _DIR / "orb_oracle.h", _DIR / "orb_pattern_data.inc"]
    if force or not _SO.exists() or any(s.stat().st_mtime > _SO.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(_DIR), "-B", "liborb_oracle.so"], check=True, capture_output=True)
    return _SO


class OFeatVec(C.Structure):
    _fields_ = [("node_id", C.c_void_p), ("node_ptr", C.c_void_p), ("index", C.c_void_p), ("n_nodes", C.c_int32)]


PAIR_PREDICATE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_SO))
        vp, i32, u32, f32, sz, u64 = C.c_void_p, C.c_int, C.c_uint32, C.c_float, C.c_size_t, C.c_uint64
        L.orbo_create.restype = vp
        L.orbo_create.argtypes = [i32, f32, i32, i32, i32, i32]
        L.orbo_destroy.argtypes = [vp]
        L.orbo_extract.restype = i32
        L.orbo_extract.argtypes = [vp, vp, i32, i32, sz, i32, i32, vp, vp, i32, vp]
        L.orbo_get_tables.restype = i32
        L.orbo_get_tables.argtypes = [vp] * 7
        L.orbo_level_size.argtypes = [vp, i32, vp, vp]
        L.orbo_level_padded.restype = vp
        L.orbo_level_padded.argtypes = [vp, i32, vp]
        L.orbo_level_blurred.restype = vp
        L.orbo_level_blurred.argtypes = [vp, i32, vp]
        L.orbo_level_candidates.restype = i32
        L.orbo_level_candidates.argtypes = [vp, i32, vp, i32]
        L.orbo_level_keypoints.restype = i32
        L.orbo_level_keypoints.argtypes = [vp, i32, vp, i32]
        L.orbo_cv_round_f.restype = i32
        L.orbo_cv_round_f.argtypes = [f32]
        L.orbo_resize_linear_u8.argtypes = [vp, i32, i32, sz, vp, i32, i32, sz]
        L.orbo_border_reflect101.argtypes = [vp, i32, i32, sz, i32]
        L.orbo_fast9_16.restype = i32
        L.orbo_fast9_16.argtypes = [vp, i32, i32, sz, i32, vp, i32]
        L.orbo_fast_score.restype = i32
        L.orbo_fast_score.argtypes = [vp, sz]
        L.orbo_gauss7_u8.argtypes = [vp, i32, i32, sz, vp, sz, i32]
        L.orbo_fast_atan2.restype = f32
        L.orbo_fast_atan2.argtypes = [f32, f32]
        L.orbo_fast_atan2_fma.restype = f32
        L.orbo_fast_atan2_fma.argtypes = [f32, f32]
        L.orbo_ic_angle.restype = f32
        L.orbo_ic_angle.argtypes = [vp, sz]
        L.orbo_sinf.restype = f32
        L.orbo_sinf.argtypes = [f32]
        L.orbo_cosf.restype = f32
        L.orbo_cosf.argtypes = [f32]
        L.orbo_check_sincos_vs_libm.restype = u64
        L.orbo_count_sincos_fma_vs_nofma.restype = u64
        L.orbo_count_sincos_fma_vs_nofma.argtypes = [u32, u32, vp]
        L.orbo_check_sincos_vs_libm.argtypes = [u32, u32, vp]
        L.orbo_orb_descriptor.argtypes = [vp, sz, f32, i32, i32, vp]
        L.orbo_distribute_octree.restype = i32
        L.orbo_distribute_octree.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i32]
        L.orbo_sort_nodes.argtypes = [vp, vp, i32, vp]
        L.orbo_descriptor_distance.restype = i32
        L.orbo_descriptor_distance.argtypes = [vp, vp]
        L.orbo_grid_create.restype = vp
        L.orbo_grid_create.argtypes = [vp, i32, f32, f32, f32, f32]
        L.orbo_grid_destroy.argtypes = [vp]
        L.orbo_grid_query.restype = i32
        L.orbo_grid_query.argtypes = [vp, f32, f32, f32, i32, i32, vp, i32]
        L.orbo_search_by_projection_mappoints.restype = i32
        L.orbo_search_by_projection_mappoints.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp,
                                                          vp, vp, f32, f32, vp]
        L.orbo_search_by_projection_frame.restype = i32
        L.orbo_search_by_projection_frame.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp,
                                                      f32, i32, i32, vp]
        L.orbo_compute_stereo_matches.restype = i32
        L.orbo_compute_stereo_matches.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, f32, f32,
                                                  vp, vp, vp, vp]
        L.orbo_knn2.argtypes = [vp, i32, vp, i32, vp, vp]
        L.orbo_search_by_projection_window.restype = i32
        L.orbo_search_by_projection_window.argtypes = [vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, f32, i32,
                                                       i32, vp]
        L.orbo_search_for_initialization.restype = i32
        L.orbo_search_for_initialization.argtypes = [vp, vp, i32, vp, vp, vp, i32, vp, i32, f32, i32, vp]
        L.orbo_search_by_bow_frame.restype = i32
        L.orbo_search_by_bow_frame.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, vp, f32, i32, vp]
        L.orbo_search_by_bow_keyframes.restype = i32
        L.orbo_search_by_bow_keyframes.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, i32, vp, f32, i32, vp]
        L.orbo_search_for_triangulation.restype = i32
        L.orbo_search_for_triangulation.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, i32, vp, i32, PAIR_PREDICATE, vp, vp]
        L.orbo_three_maxima.argtypes = [vp, i32, vp, vp, vp]
        L.orbo_fuse_search.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp]
        L.orbo_distinctive_descriptors.argtypes = [vp, vp, i32, vp]
        L.orbo_bow_transform.argtypes = [vp, vp, vp, vp, i32, i32, vp, i32, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleExtractor:
    """ORBextractor restated on the CPU (the parity oracle)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, flags=FLAG_DESC_FMA):
        self.L = lib()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = self.L.orbo_create(nfeatures, scale_factor, nlevels, ini_th, min_th, flags)
        if not self.h:
            raise ValueError("bad extractor parameters")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orbo_destroy(self.h)
            self.h = None

    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        quota = np.zeros(n, np.int32)
        umax = np.zeros(16, np.int32)
        self.L.orbo_get_tables(self.h, _p(sc), _p(isc), _p(s2), _p(is2), _p(quota), _p(umax))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, quota=quota, umax=umax)

    def extract(self, img: np.ndarray, lap=(0, 0)):
        """Returns (mono_index, keypoints[KP_DTYPE], descriptors[N,32])."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        cap = self.nfeatures * 4 + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        r = self.L.orbo_extract(self.h, _p(img), w, h, img.strides[0], lap[0], lap[1], _p(kps), _p(desc), cap,
                                C.byref(n))
        if r < 0:
            return r, kps[:0], desc[:0]
        return r, kps[:n.value].copy(), desc[:n.value].copy()

    def level_size(self, lvl):
        w, h = C.c_int(0), C.c_int(0)
        self.L.orbo_level_size(self.h, lvl, C.byref(w), C.byref(h))
        return w.value, h.value

    def level_padded(self, lvl) -> np.ndarray:
        w, h = self.level_size(lvl)
        st = C.c_size_t(0)
        p = self.L.orbo_level_padded(self.h, lvl, C.byref(st))
        buf = (C.c_uint8 * (st.value * (h + 38))).from_address(p)
        return np.frombuffer(buf, np.uint8).reshape(h + 38, st.value)[:, :w + 38].copy()

    def level_blurred(self, lvl):
        w, h = self.level_size(lvl)
        st = C.c_size_t(0)
        p = self.L.orbo_level_blurred(self.h, lvl, C.byref(st))
        if not p:
            return None
        buf = (C.c_uint8 * (st.value * h)).from_address(p)
        return np.frombuffer(buf, np.uint8).reshape(h, st.value)[:, :w].copy()

    def level_candidates(self, lvl) -> np.ndarray:
        n = self.L.orbo_level_candidates(self.h, lvl, None, 0)
        out = np.zeros(n, KP_DTYPE)
        self.L.orbo_level_candidates(self.h, lvl, _p(out), n)
        return out

    def level_keypoints(self, lvl) -> np.ndarray:
        n = self.L.orbo_level_keypoints(self.h, lvl, None, 0)
        out = np.zeros(n, KP_DTYPE)
        self.L.orbo_level_keypoints(self.h, lvl, _p(out), n)
        return out


# ---- primitive wrappers -------------------------------------------------------------------------
def resize_linear(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orbo_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dst.strides[0])
    return dst


def fast9_16(img: np.ndarray, threshold: int) -> np.ndarray:
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    out = np.zeros(cap, KP_DTYPE)
    n = lib().orbo_fast9_16(_p(img), img.shape[1], img.shape[0], img.strides[0], threshold, _p(out), cap)
    return out[:n].copy()


def fast_score_map(img: np.ndarray) -> np.ndarray:
    """score(p) for every pixel at least 3 px from the border (others 0); negative clamps to 0."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.int32)
    L = lib()
    base = img.ctypes.data
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            out[y, x] = L.orbo_fast_score(C.c_void_p(base + y * img.strides[0] + x), img.strides[0])
    return out


def gauss7(img: np.ndarray, ocv440: bool = False) -> np.ndarray:
    img = np.ascontiguousarray(img, np.uint8)
    dst = np.zeros_like(img)
    lib().orbo_gauss7_u8(_p(img), img.shape[1], img.shape[0], img.strides[0], _p(dst), dst.strides[0], int(ocv440))
    return dst


def fast_atan2(y: float, x: float, fma: bool = False) -> float:
    return lib().orbo_fast_atan2_fma(y, x) if fma else lib().orbo_fast_atan2(y, x)


def ic_angle(img: np.ndarray, x: int, y: int) -> float:
    img = np.ascontiguousarray(img, np.uint8)
    return lib().orbo_ic_angle(C.c_void_p(img.ctypes.data + y * img.strides[0] + x), img.strides[0])


def orb_descriptor(img: np.ndarray, x: int, y: int, angle_deg: float, fma: bool = True, libm: bool = False):
    img = np.ascontiguousarray(img, np.uint8)
    d = np.zeros(32, np.uint8)
    lib().orbo_orb_descriptor(C.c_void_p(img.ctypes.data + y * img.strides[0] + x), img.strides[0], angle_deg,
                              int(fma), int(libm), _p(d))
    return d


def distribute_octree(cands: np.ndarray, minX, maxX, minY, maxY, N) -> np.ndarray:
    cands = np.ascontiguousarray(cands, KP_DTYPE)
    out = np.zeros(len(cands) + 8, KP_DTYPE)
    n = lib().orbo_distribute_octree(_p(cands), len(cands), minX, maxX, minY, maxY, N, _p(out), len(out))
    return out[:n].copy()


def sort_nodes(count: np.ndarray, ulx: np.ndarray) -> np.ndarray:
    count = np.ascontiguousarray(count, np.int32)
    ulx = np.ascontiguousarray(ulx, np.int32)
    perm = np.zeros(len(count), np.int32)
    lib().orbo_sort_nodes(_p(count), _p(ulx), len(count), _p(perm))
    return perm


def descriptor_distance(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orbo_descriptor_distance(_p(a), _p(b))


class OracleGrid:
    def __init__(self, kps_un: np.ndarray, minx, maxx, miny, maxy):
        self.kps = np.ascontiguousarray(kps_un, KP_DTYPE)
        self.L = lib()
        self.h = self.L.orbo_grid_create(_p(self.kps), len(self.kps), minx, maxx, miny, maxy)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orbo_grid_destroy(self.h)
            self.h = None

    def query(self, x, y, r, min_level=-1, max_level=-1) -> np.ndarray:
        out = np.zeros(len(self.kps) + 1, np.int32)
        n = self.L.orbo_grid_query(self.h, x, y, r, min_level, max_level, _p(out), len(out))
        return out[:n].copy()


def search_by_projection_mappoints(grid: OracleGrid, frame_desc, scale_factors, mp, th, nnratio, u_right=None,
                                   occupied=None):
    """mp: dict(proj_x, proj_y, proj_xr, level, view_cos, desc, in_view, has_obs). Returns (nmatches, frame_match)."""
    nF = len(grid.kps)
    frame_desc = np.ascontiguousarray(frame_desc, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    fm = np.full(nF, -1, np.int32)
    a = {k: np.ascontiguousarray(v) for k, v in mp.items()}
    n = lib().orbo_search_by_projection_mappoints(
        grid.h, _p(grid.kps), _p(frame_desc), nF, _p(sf), _p(u_right), _p(occupied), len(a["proj_x"]),
        _p(a["proj_x"].astype(np.float32)), _p(a["proj_y"].astype(np.float32)),
        _p(a["proj_xr"].astype(np.float32)), _p(a["level"].astype(np.int32)), _p(a["view_cos"].astype(np.float32)),
        _p(a["desc"].astype(np.uint8)), _p(a["in_view"].astype(np.uint8)), _p(a["has_obs"].astype(np.uint8)),
        th, nnratio, _p(fm))
    return n, fm


def search_by_projection_frame(grid: OracleGrid, cur_desc, scale_factors, q, th, mode=0, check_orientation=True,
                               cur_u_right=None, cur_occupied=None):
    nC = len(grid.kps)
    cur_desc = np.ascontiguousarray(cur_desc, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    cm = np.full(nC, -1, np.int32)
    a = {k: np.ascontiguousarray(v) for k, v in q.items()}
    n = lib().orbo_search_by_projection_frame(
        grid.h, _p(grid.kps), _p(cur_desc), nC, _p(sf), _p(cur_u_right), _p(cur_occupied), len(a["u"]),
        _p(a["u"].astype(np.float32)), _p(a["v"].astype(np.float32)), _p(a["ur"].astype(np.float32)),
        _p(a["octave"].astype(np.int32)), _p(a["angle"].astype(np.float32)), _p(a["desc"].astype(np.uint8)),
        _p(a["has_obs"].astype(np.uint8)), th, mode, int(check_orientation), _p(cm))
    return n, cm


def knn2(q: np.ndarray, t: np.ndarray):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    idx = np.zeros((len(q), 2), np.int32)
    dist = np.zeros((len(q), 2), np.int32)
    lib().orbo_knn2(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist))
    return idx, dist


TRIANGULATE_FN = C.CFUNCTYPE(C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_float))


def triangulate_callback(fn):
    """fn(i_left, i_right, sigma1, sigma2) -> (depth, (x, y, z)): the stand-in for KannalaBrandt8::TriangulateMatches as a C callback."""
    def cb(_ctx, il, ir, s1, s2, p3d):
        d, p = fn(il, ir, s1, s2)
        p3d[0], p3d[1], p3d[2] = float(p[0]), float(p[1]), float(p[2])
        return float(d)
    return TRIANGULATE_FN(cb)


def stereo_fisheye_matches(kl, dl, mono_left, kr, dr, mono_right, level_sigma2, triangulate):
    """Frame::ComputeStereoFishEyeMatches (Frame.cc:1126-1166).  Returns (nMatches, descMatches, l2r, r2l, depth, u_right, p3d[n_left, 3])."""
    kl, kr = np.ascontiguousarray(kl, KP_DTYPE), np.ascontiguousarray(kr, KP_DTYPE)
    dl, dr = np.ascontiguousarray(dl, np.uint8), np.ascontiguousarray(dr, np.uint8)
    s2 = np.ascontiguousarray(level_sigma2, np.float32)
    l2r, r2l = np.zeros(len(kl), np.int32), np.zeros(len(kr), np.int32)
    depth, ur, p3d = np.zeros(len(kl), np.float32), np.zeros(len(kl), np.float32), np.zeros((len(kl), 3), np.float32)
    nd = C.c_int(0)
    cb = triangulate_callback(triangulate)
    f = lib().orbo_stereo_fisheye_matches
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, TRIANGULATE_FN, C.c_void_p] + [C.c_void_p] * 6
    n = f(_p(kl), _p(dl), len(kl), int(mono_left), _p(kr), _p(dr), len(kr), int(mono_right), _p(s2), cb, None,
          _p(l2r), _p(r2l), _p(depth), _p(ur), _p(p3d), C.cast(C.byref(nd), C.c_void_p))
    return n, nd.value, l2r, r2l, depth, ur, p3d


def compute_stereo_matches(kl, dl, kr, dr, scale, inv_scale, pyr_l, pyr_r, bf, b):
    """pyr_l / pyr_r: lists of contiguous uint8 level images (ROI, no ring)."""
    kl = np.ascontiguousarray(kl, KP_DTYPE)
    kr = np.ascontiguousarray(kr, KP_DTYPE)
    dl = np.ascontiguousarray(dl, np.uint8)
    dr = np.ascontiguousarray(dr, np.uint8)
    nl = len(pyr_l)
    pl = (C.c_void_p * nl)(*[p.ctypes.data for p in pyr_l])
    pr = (C.c_void_p * nl)(*[p.ctypes.data for p in pyr_r])
    pw = np.array([p.shape[1] for p in pyr_l], np.int32)
    ph = np.array([p.shape[0] for p in pyr_l], np.int32)
    ps = np.array([p.strides[0] for p in pyr_l], np.uint64)
    N = len(kl)
    ur = np.zeros(N, np.float32)
    depth = np.zeros(N, np.float32)
    bi = np.zeros(N, np.int32)
    bd = np.zeros(N, np.int32)
    sc = np.ascontiguousarray(scale, np.float32)
    isc = np.ascontiguousarray(inv_scale, np.float32)
    n = lib().orbo_compute_stereo_matches(_p(kl), _p(dl), N, _p(kr), _p(dr), len(kr), _p(sc), _p(isc), nl,
                                          C.cast(pl, C.c_void_p), C.cast(pr, C.c_void_p), _p(pw), _p(ph), _p(ps),
                                          bf, b, _p(ur), _p(depth), _p(bi), _p(bd))
    return n, ur, depth, bi, bd


def three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, np.int32)
    a, b, c = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    lib().orbo_three_maxima(_p(sizes), len(sizes), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def _fv(fv):
    """fv: any object with node_id / node_ptr / index numpy arrays (e.g. orb_slam3_amd.FeatureVector)."""
    return OFeatVec(fv.node_id.ctypes.data, fv.node_ptr.ctypes.data, fv.index.ctypes.data, len(fv.node_id))


def search_by_projection_window(grid: OracleGrid, desc, q, max_dist, check_orientation, level_gate_in_loop=False, occupied=None):
    n = len(grid.kps)
    desc = np.ascontiguousarray(desc, np.uint8)
    match = np.full(n, -1, np.int32)
    a = {k: np.ascontiguousarray(v) for k, v in q.items()}
    ang = a["angle"].astype(np.float32) if "angle" in a else np.zeros(len(a["x"]), np.float32)
    ho = a["has_obs"].astype(np.uint8) if "has_obs" in a else None
    r = lib().orbo_search_by_projection_window(
        grid.h, _p(grid.kps), _p(desc), n, _p(occupied), len(a["x"]), _p(a["x"].astype(np.float32)),
        _p(a["y"].astype(np.float32)), _p(a["r"].astype(np.float32)), _p(a["min_level"].astype(np.int32)),
        _p(a["max_level"].astype(np.int32)), _p(ang), _p(a["desc"].astype(np.uint8)), _p(ho), max_dist,
        int(check_orientation), int(level_gate_in_loop), _p(match))
    return r, match


def search_for_initialization(kps1, desc1, grid2: OracleGrid, desc2, prev_matched, window_size, nnratio, check_orientation):
    kps1 = np.ascontiguousarray(kps1, KP_DTYPE)
    desc1 = np.ascontiguousarray(desc1, np.uint8)
    desc2 = np.ascontiguousarray(desc2, np.uint8)
    assert prev_matched.dtype == np.float32 and prev_matched.flags.c_contiguous
    m12 = np.full(len(kps1), -1, np.int32)
    n = lib().orbo_search_for_initialization(_p(kps1), _p(desc1), len(kps1), grid2.h, _p(grid2.kps), _p(desc2), len(grid2.kps),
                                             _p(prev_matched), window_size, nnratio, int(check_orientation), _p(m12))
    return n, m12


def search_by_bow_frame(kf_desc, kf_angle, kf_valid, kf_fv, f_desc, f_angle, f_fv, nnratio, check_orientation):
    kd, fd = np.ascontiguousarray(kf_desc, np.uint8), np.ascontiguousarray(f_desc, np.uint8)
    ka, fa = np.ascontiguousarray(kf_angle, np.float32), np.ascontiguousarray(f_angle, np.float32)
    kv = np.ascontiguousarray(kf_valid, np.uint8)
    a, b = _fv(kf_fv), _fv(f_fv)
    fm = np.full(len(fd), -1, np.int32)
    n = lib().orbo_search_by_bow_frame(_p(kd), _p(ka), _p(kv), len(kd), C.byref(a), _p(fd), _p(fa), len(fd), C.byref(b), nnratio,
                                       int(check_orientation), _p(fm))
    return n, fm


def search_by_bow_keyframes(d1, a1, v1, fv1, d2, a2, v2, fv2, nnratio, check_orientation):
    d1, d2 = np.ascontiguousarray(d1, np.uint8), np.ascontiguousarray(d2, np.uint8)
    a1, a2 = np.ascontiguousarray(a1, np.float32), np.ascontiguousarray(a2, np.float32)
    v1, v2 = np.ascontiguousarray(v1, np.uint8), np.ascontiguousarray(v2, np.uint8)
    a, b = _fv(fv1), _fv(fv2)
    m12 = np.full(len(d1), -1, np.int32)
    n = lib().orbo_search_by_bow_keyframes(_p(d1), _p(a1), _p(v1), len(d1), C.byref(a), _p(d2), _p(a2), _p(v2), len(d2),
                                           C.byref(b), nnratio, int(check_orientation), _p(m12))
    return n, m12


def search_for_triangulation(d1, a1, s1, fv1, d2, a2, s2, fv2, check_orientation, pair_ok=None):
    d1, d2 = np.ascontiguousarray(d1, np.uint8), np.ascontiguousarray(d2, np.uint8)
    a1, a2 = np.ascontiguousarray(a1, np.float32), np.ascontiguousarray(a2, np.float32)
    s1, s2 = np.ascontiguousarray(s1, np.uint8), np.ascontiguousarray(s2, np.uint8)
    a, b = _fv(fv1), _fv(fv2)
    m12 = np.full(len(d1), -1, np.int32)
    cb = PAIR_PREDICATE((lambda user, i, j: int(bool(pair_ok(i, j)))) if pair_ok else 0)
    n = lib().orbo_search_for_triangulation(_p(d1), _p(a1), _p(s1), len(d1), C.byref(a), _p(d2), _p(a2), _p(s2), len(d2),
                                            C.byref(b), int(check_orientation), cb, None, _p(m12))
    return n, m12


def bow_transform(child_ptr, child_idx, node_desc, word_id, L, levelsup, desc):
    cp, ci = np.ascontiguousarray(child_ptr, np.int32), np.ascontiguousarray(child_idx, np.int32)
    nd, wi = np.ascontiguousarray(node_desc, np.uint8), np.ascontiguousarray(word_id, np.int32)
    d = np.ascontiguousarray(desc, np.uint8)
    word = np.zeros(len(d), np.int32)
    node = np.zeros(len(d), np.int32)
    lib().orbo_bow_transform(_p(cp), _p(ci), _p(nd), _p(wi), L, levelsup, _p(d), len(d), _p(word), _p(node))
    return word, node


def distinctive_descriptors(desc, set_ptr):
    d = np.ascontiguousarray(desc, np.uint8)
    sp = np.ascontiguousarray(set_ptr, np.int32)
    out = np.zeros(len(sp) - 1, np.int32)
    lib().orbo_distinctive_descriptors(_p(d), _p(sp), len(sp) - 1, _p(out))
    return out


def fuse_search(grid: OracleGrid, desc, u_right, inv_sigma2, q, fma=True):
    desc = np.ascontiguousarray(desc, np.uint8)
    a = {k: np.ascontiguousarray(v) for k, v in q.items()}
    nq = len(a["u"])
    bi = np.zeros(nq, np.int32)
    bd = np.zeros(nq, np.int32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    isg = None if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float32)
    lib().orbo_fuse_search(grid.h, _p(grid.kps), _p(desc), len(grid.kps), _p(ur), _p(isg), nq, _p(a["u"].astype(np.float32)),
                           _p(a["v"].astype(np.float32)), _p(a["ur"].astype(np.float32)), _p(a["r"].astype(np.float32)),
                           _p(a["level"].astype(np.int32)), _p(a["desc"].astype(np.uint8)), int(fma), _p(bi), _p(bd))
    return bi, bd


def epipolar_pinhole(F12, x1, y1, x2, y2, unc, fma=True):
    """Pinhole::epipolarConstrain on a given F12 for arrays of keypoint pairs -> uint8 verdicts."""
    F = np.ascontiguousarray(F12, np.float32).ravel()
    L = lib()
    L.orbo_epipolar_pinhole.restype = C.c_int
    f = C.c_float
    return np.array([L.orbo_epipolar_pinhole(_p(F), f(a), f(b), f(c), f(d), f(u), int(fma)) for a, b, c, d, u in zip(x1, y1, x2, y2, unc)],
                    np.uint8)


def search_for_triangulation_pinhole(k1, d1, s1, ur1, fv1, k2, d2, s2, ur2, fv2, scale2, sigma2_2, F12, ep, coarse, check_orientation,
                                     fma=True):
    k1, k2 = np.ascontiguousarray(k1, KP_DTYPE), np.ascontiguousarray(k2, KP_DTYPE)
    d1, d2 = np.ascontiguousarray(d1, np.uint8), np.ascontiguousarray(d2, np.uint8)
    s1, s2 = np.ascontiguousarray(s1, np.uint8), np.ascontiguousarray(s2, np.uint8)
    ur1 = None if ur1 is None else np.ascontiguousarray(ur1, np.float32)
    ur2 = None if ur2 is None else np.ascontiguousarray(ur2, np.float32)
    sc, sg, F = (np.ascontiguousarray(x, np.float32) for x in (scale2, sigma2_2, np.asarray(F12).ravel()))
    a, b = _fv(fv1), _fv(fv2)
    m12 = np.full(len(k1), -1, np.int32)
    L = lib()
    L.orbo_search_for_triangulation_pinhole.restype = C.c_int
    n = L.orbo_search_for_triangulation_pinhole(_p(k1), _p(d1), _p(s1), _p(ur1), len(k1), C.byref(a), _p(k2), _p(d2), _p(s2), _p(ur2),
                                                len(k2), C.byref(b), _p(sc), _p(sg), _p(F), C.c_float(ep[0]), C.c_float(ep[1]),
                                                int(coarse), int(check_orientation), int(fma), _p(m12))
    return n, m12


def search_for_triangulation_kb8(k1, n_left1, d1, s1, fv1, k2, n_left2, d2, s2, fv2, sigma2_1, sigma2_2, cam1, cam2, R12, t12, coarse, check_orientation):
    """M7 between key frames of a fisheye rig with KannalaBrandt8::epipolarConstrain as the lazily evaluated gate (ORBmatcher.cc:1036-1072)."""
    f32 = np.float32
    k1, k2 = np.ascontiguousarray(k1, KP_DTYPE), np.ascontiguousarray(k2, KP_DTYPE)
    d1, d2 = np.ascontiguousarray(d1, np.uint8), np.ascontiguousarray(d2, np.uint8)
    s1, s2 = np.ascontiguousarray(s1, np.uint8), np.ascontiguousarray(s2, np.uint8)
    sg1, sg2, c1, c2, R, t = [np.ascontiguousarray(x, f32).ravel() for x in (sigma2_1, sigma2_2, cam1, cam2, R12, t12)]
    a, b = _fv(fv1), _fv(fv2)
    m12 = np.full(len(k1), -1, np.int32)
    L = lib()
    L.orbo_search_for_triangulation_kb8.restype = C.c_int
    n = L.orbo_search_for_triangulation_kb8(_p(k1), int(n_left1), _p(d1), _p(s1), len(k1), C.byref(a), _p(k2), int(n_left2), _p(d2), _p(s2), len(k2), C.byref(b),
                                            _p(sg1), _p(sg2), _p(c1), _p(c2), _p(R), _p(t), int(coarse), int(check_orientation), _p(m12))
    return n, m12


def search_by_projection_mappoints_fisheye(grid_left: OracleGrid, grid_right: OracleGrid, desc, scale_factors, l2r, r2l, mp, th, nnratio,
                                           occupied=None):
    """mp: in_view, proj_x, proj_y, level, view_cos, in_view_r, proj_xr, proj_yr, level_r, view_cos_r, desc, has_obs."""
    nl, nr = len(grid_left.kps), len(grid_right.kps)
    desc = np.ascontiguousarray(desc, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    l2r, r2l = np.ascontiguousarray(l2r, np.int32), np.ascontiguousarray(r2l, np.int32)
    occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    fm = np.full(nl + nr, -1, np.int32)
    f32, i32, u8 = np.float32, np.int32, np.uint8
    a = [np.ascontiguousarray(mp[k], t) for k, t in (("in_view", u8), ("proj_x", f32), ("proj_y", f32), ("level", i32), ("view_cos", f32),
                                                     ("in_view_r", u8), ("proj_xr", f32), ("proj_yr", f32), ("level_r", i32),
                                                     ("view_cos_r", f32), ("desc", u8), ("has_obs", u8))]
    L = lib()
    L.orbo_search_by_projection_mappoints_fisheye.restype = C.c_int
    n = L.orbo_search_by_projection_mappoints_fisheye(C.c_void_p(grid_left.h), C.c_void_p(grid_right.h), _p(grid_left.kps), nl, _p(grid_right.kps), nr, _p(desc), _p(sf),
                                                      _p(l2r), _p(r2l), _p(occ), len(a[0]), *[_p(x) for x in a], C.c_float(th),
                                                      C.c_float(nnratio), _p(fm))
    return n, fm


def search_by_projection_frame_fisheye(grid_left: OracleGrid, grid_right: OracleGrid, desc, scale_factors, q, th, mode=0, check_orientation=True,
                                       occupied=None):
    """q: u, v, xr, yr (projection into the right camera), octave, angle, desc, has_obs."""
    nl, nr = len(grid_left.kps), len(grid_right.kps)
    desc = np.ascontiguousarray(desc, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    cm = np.full(nl + nr, -1, np.int32)
    f32, i32, u8 = np.float32, np.int32, np.uint8
    a = [np.ascontiguousarray(q[k], t) for k, t in (("u", f32), ("v", f32), ("xr", f32), ("yr", f32), ("octave", i32), ("angle", f32),
                                                    ("desc", u8), ("has_obs", u8))]
    L = lib()
    L.orbo_search_by_projection_frame_fisheye.restype = C.c_int
    n = L.orbo_search_by_projection_frame_fisheye(C.c_void_p(grid_left.h), C.c_void_p(grid_right.h), _p(grid_left.kps), nl, _p(grid_right.kps), nr,
                                                  _p(desc), _p(sf), _p(occ), len(a[0]), *[_p(x) for x in a], C.c_float(th), int(mode),
                                                  int(check_orientation), _p(cm))
    return n, cm


def search_by_bow_frame_fisheye(kf_desc, kf_angle, kf_valid, kf_fv, f_desc, f_angle, n_f_left, f_fv, nnratio, check_orientation):
    kd, fd = np.ascontiguousarray(kf_desc, np.uint8), np.ascontiguousarray(f_desc, np.uint8)
    ka, fa = np.ascontiguousarray(kf_angle, np.float32), np.ascontiguousarray(f_angle, np.float32)
    kv = np.ascontiguousarray(kf_valid, np.uint8)
    a, b = _fv(kf_fv), _fv(f_fv)
    fm = np.full(len(fd), -1, np.int32)
    L = lib()
    L.orbo_search_by_bow_frame_fisheye.restype = C.c_int
    n = L.orbo_search_by_bow_frame_fisheye(_p(kd), _p(ka), _p(kv), len(kd), C.byref(a), _p(fd), _p(fa), len(fd), int(n_f_left), C.byref(b),
                                           C.c_float(nnratio), int(check_orientation), _p(fm))
    return n, fm


def is_in_frustum(Rcw, tcw, Ow, cam, bounds, log_scale_factor, nlevels, cos_limit, pos, normal, min_dist, max_dist):
    """Frame::isInFrustum (Nleft == -1) for n map points.  cam = (fx, fy, cx, cy, mbf).  Returns dict(in_view, proj_x, proj_y, proj_xr,
    depth, level, view_cos)."""
    f32 = np.float32
    R, t, O, b = [np.ascontiguousarray(x, f32).ravel() for x in (Rcw, tcw, Ow, bounds)]
    P, Nn = np.ascontiguousarray(pos, f32).reshape(-1, 3), np.ascontiguousarray(normal, f32).reshape(-1, 3)
    mn, mx = np.ascontiguousarray(min_dist, f32), np.ascontiguousarray(max_dist, f32)
    n = len(P)
    out = dict(in_view=np.zeros(n, np.uint8), proj_x=np.zeros(n, f32), proj_y=np.zeros(n, f32), proj_xr=np.zeros(n, f32),
               depth=np.zeros(n, f32), level=np.zeros(n, np.int32), view_cos=np.zeros(n, f32))
    L = lib()
    L.orbo_is_in_frustum.restype = None
    L.orbo_is_in_frustum(_p(R), _p(t), _p(O), C.c_float(cam[0]), C.c_float(cam[1]), C.c_float(cam[2]), C.c_float(cam[3]), C.c_float(cam[4]), _p(b),
                         C.c_float(log_scale_factor), int(nlevels), C.c_float(cos_limit), n, _p(P), _p(Nn), _p(mn), _p(mx),
                         *[_p(out[k]) for k in ("in_view", "proj_x", "proj_y", "proj_xr", "depth", "level", "view_cos")])
    return out


def is_in_frustum_checks(view, bounds, log_scale_factor, nlevels, cos_limit, pos, normal, min_dist, max_dist):
    """Frame::isInFrustumChecks for one camera of a fisheye rig.  view = (R, t, twc, params8) as Frame.cc:1172-1186 compute them.
    Returns dict(in_view, proj_x, proj_y, depth, level, view_cos)."""
    f32 = np.float32
    R, t, w, prm = [np.ascontiguousarray(x, f32).ravel() for x in view]
    b = np.ascontiguousarray(bounds, f32).ravel()
    P, Nn = np.ascontiguousarray(pos, f32).reshape(-1, 3), np.ascontiguousarray(normal, f32).reshape(-1, 3)
    mn, mx = np.ascontiguousarray(min_dist, f32), np.ascontiguousarray(max_dist, f32)
    n = len(P)
    out = dict(in_view=np.zeros(n, np.uint8), proj_x=np.zeros(n, f32), proj_y=np.zeros(n, f32), depth=np.zeros(n, f32),
               level=np.zeros(n, np.int32), view_cos=np.zeros(n, f32))
    L = lib()
    L.orbo_is_in_frustum_checks.restype = None
    L.orbo_is_in_frustum_checks(_p(R), _p(t), _p(w), _p(prm), _p(b), C.c_float(log_scale_factor), int(nlevels), C.c_float(cos_limit), n, _p(P), _p(Nn),
                                _p(mn), _p(mx), *[_p(out[k]) for k in ("in_view", "proj_x", "proj_y", "depth", "level", "view_cos")])
    return out


def kb8_epipolar_constrain(cam1, cam2, xy1, xy2, R12, t12, sigma1, sigma2):
    """KannalaBrandt8::epipolarConstrain / TriangulateMatches for n keypoint pairs.  Returns (ok uint8[n], TriangulateMatches value float32[n])."""
    f32 = np.float32
    c1, c2, R, t = [np.ascontiguousarray(x, f32).ravel() for x in (cam1, cam2, R12, t12)]
    a, b = np.ascontiguousarray(xy1, f32).reshape(-1, 2), np.ascontiguousarray(xy2, f32).reshape(-1, 2)
    s1, s2 = np.ascontiguousarray(sigma1, f32), np.ascontiguousarray(sigma2, f32)
    n = len(a)
    ok, val = np.zeros(n, np.uint8), np.zeros(n, f32)
    L = lib()
    L.orbo_kb8_epipolar_constrain.restype = None
    L.orbo_kb8_epipolar_constrain(_p(c1), _p(c2), n, _p(a), _p(b), _p(R), _p(t), _p(s1), _p(s2), _p(ok), _p(val))
    return ok, val


def kb8_unproject(cam, xy):
    f32 = np.float32
    c = np.ascontiguousarray(cam, f32).ravel()
    a = np.ascontiguousarray(xy, f32).reshape(-1, 2)
    out = np.zeros((len(a), 3), f32)
    L = lib()
    L.orbo_kb8_unproject.restype = None
    for i in range(len(a)):
        L.orbo_kb8_unproject(_p(c), C.c_float(a[i, 0]), C.c_float(a[i, 1]), _p(out[i]))
    return out


def undistort_points(xy, cam, dist):
    """cv::undistortPoints(xy, K, dist, R=I, P=K) [OCV-recalled].  cam = (fx, fy, cx, cy), dist = (k1, k2, p1, p2[, k3])."""
    a = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    o = np.zeros_like(a)
    d = list(dist) + [0.0] * (5 - len(dist))
    L = lib()
    L.orbo_undistort_points.restype = None
    L.orbo_undistort_points(len(a), _p(a), *[C.c_float(x) for x in cam[:4]], *[C.c_float(x) for x in d], _p(o))
    return o


def image_bounds(width, height, cam, dist):
    b = np.zeros(4, np.float32)
    d = list(dist) + [0.0] * (5 - len(dist))
    L = lib()
    L.orbo_image_bounds.restype = None
    L.orbo_image_bounds(int(width), int(height), *[C.c_float(x) for x in cam[:4]], *[C.c_float(x) for x in d], _p(b))
    return b
