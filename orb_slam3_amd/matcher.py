"""Host-side mirror of ORB_SLAM3::ORBmatcher (/root/reference/include/ORBmatcher.h:36-103) over the C ABI.

The reference matchers walk Frame / KeyFrame / MapPoint pointer graphs; at this boundary those are passed
flattened (numpy arrays).  `FrameView` carries what the matchers read of a Frame: undistorted keypoints,
descriptors, image bounds (for the 64x48 grid) and the per-level scale factors.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, FrameDesc, check, ptr


@dataclass
class FrameView:
    keypoints_un: np.ndarray          # KP_DTYPE [N]   (Frame::mvKeysUn)
    descriptors: np.ndarray           # uint8 [N,32]   (Frame::mDescriptors)
    min_x: float
    max_x: float
    min_y: float
    max_y: float
    scale_factors: np.ndarray         # float32 [nlevels] (Frame::mvScaleFactors)
    u_right: np.ndarray | None = None  # float32 [N] (Frame::mvuRight) or None for mono

    def c_struct(self):
        self.keypoints_un = np.ascontiguousarray(self.keypoints_un, KP_DTYPE)
        self.descriptors = np.ascontiguousarray(self.descriptors, np.uint8)
        self.scale_factors = np.ascontiguousarray(self.scale_factors, np.float32)
        if self.u_right is not None:
            self.u_right = np.ascontiguousarray(self.u_right, np.float32)
        return FrameDesc(self.keypoints_un.ctypes.data, self.descriptors.ctypes.data, len(self.keypoints_un),
                         self.min_x, self.max_x, self.min_y, self.max_y, self.scale_factors.ctypes.data,
                         len(self.scale_factors), None if self.u_right is None else self.u_right.ctypes.data)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, np.float32)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, np.int32)


def _u8(a):
    return None if a is None else np.ascontiguousarray(a, np.uint8)


class ORBmatcher:
    TH_LOW = _lib.TH_LOW
    TH_HIGH = _lib.TH_HIGH
    HISTO_LENGTH = _lib.HISTO_LENGTH

    def __init__(self, nnratio: float = 0.6, checkOri: bool = True, device: int = 0):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        check(self._L.orbx_matcher_create(device, C.byref(self._h)), "orbx_matcher_create")
        self.mfNNratio = nnratio
        self.mbCheckOrientation = checkOri

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orbx_matcher_destroy(self._h)
            self._h = None

    # ---- DescriptorDistance over candidate lists (ORBmatcher.cc:2058-2074) ----
    def hamming_csr(self, q_desc, t_desc, row_ptr, cand):
        q, t, rp, cd = _u8(q_desc), _u8(t_desc), _i32(row_ptr), _i32(cand)
        out = np.zeros(len(cd), np.uint16)
        check(self._L.orbx_hamming_csr(self._h, ptr(q), len(q), ptr(t), len(t), ptr(rp), ptr(cd), ptr(out)),
              "orbx_hamming_csr")
        return out

    def hamming_best2_csr(self, q_desc, t_desc, row_ptr, cand):
        q, t, rp, cd = _u8(q_desc), _u8(t_desc), _i32(row_ptr), _i32(cand)
        o = [np.zeros(len(q), np.int32) for _ in range(4)]
        check(self._L.orbx_hamming_best2_csr(self._h, ptr(q), len(q), ptr(t), len(t), ptr(rp), ptr(cd),
                                             *[ptr(a) for a in o]), "orbx_hamming_best2_csr")
        return tuple(o)

    def knn2(self, q_desc, t_desc):
        """cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, k=2) (Frame.cc:1144)."""
        q, t = _u8(q_desc), _u8(t_desc)
        idx = np.zeros((len(q), 2), np.int32)
        dist = np.zeros((len(q), 2), np.int32)
        check(self._L.orbx_knn2(self._h, ptr(q), len(q), ptr(t), len(t), ptr(idx), ptr(dist)), "orbx_knn2")
        return idx, dist

    def stereo_rowband(self, kl, dl, kr, dr, scale_factors, n_rows, min_d, max_d):
        kl = np.ascontiguousarray(kl, KP_DTYPE)
        kr = np.ascontiguousarray(kr, KP_DTYPE)
        dl, dr, sf = _u8(dl), _u8(dr), _f32(scale_factors)
        bi = np.zeros(len(kl), np.int32)
        bd = np.zeros(len(kl), np.int32)
        check(self._L.orbx_stereo_rowband(self._h, ptr(kl), ptr(dl), len(kl), ptr(kr), ptr(dr), len(kr), ptr(sf),
                                          len(sf), n_rows, min_d, max_d, ptr(bi), ptr(bd)), "orbx_stereo_rowband")
        return bi, bd

    def ComputeStereoMatches(self, kl, dl, kr, dr, scale_factors, inv_scale_factors, pyr_left, pyr_right, bf, b):
        """Frame::ComputeStereoMatches (Frame.cc:811-981). pyr_*: lists of level ROI arrays (mvImagePyramid)."""
        kl = np.ascontiguousarray(kl, KP_DTYPE)
        kr = np.ascontiguousarray(kr, KP_DTYPE)
        dl, dr, sf, isf = _u8(dl), _u8(dr), _f32(scale_factors), _f32(inv_scale_factors)
        nl = len(pyr_left)
        pl = (C.c_void_p * nl)(*[p.ctypes.data for p in pyr_left])
        pr = (C.c_void_p * nl)(*[p.ctypes.data for p in pyr_right])
        pw = np.array([p.shape[1] for p in pyr_left], np.int32)
        ph = np.array([p.shape[0] for p in pyr_left], np.int32)
        ps = np.array([p.strides[0] for p in pyr_left], np.uint64)
        for a, b_ in zip(pyr_left, pyr_right):
            assert a.strides == b_.strides and a.strides[1] == 1
        ur = np.zeros(len(kl), np.float32)
        depth = np.zeros(len(kl), np.float32)
        n = check(self._L.orbx_compute_stereo_matches(self._h, ptr(kl), ptr(dl), len(kl), ptr(kr), ptr(dr), len(kr),
                                                      ptr(sf), ptr(isf), nl, C.cast(pl, C.c_void_p),
                                                      C.cast(pr, C.c_void_p), ptr(pw), ptr(ph), ptr(ps), bf, b,
                                                      ptr(ur), ptr(depth)), "orbx_compute_stereo_matches")
        return n, ur, depth

    # ---- SearchByProjection(Frame&, vector<MapPoint*>&, th, ...) (ORBmatcher.cc:43-213) ----
    def SearchByProjection(self, F: FrameView, mp: dict, th: float = 3.0, frame_occupied=None):
        """mp: proj_x, proj_y, proj_xr, level, view_cos, desc, in_view, has_obs.  Returns (nmatches, frame_match)."""
        fd = F.c_struct()
        n_mp = len(mp["proj_x"])
        fm = np.full(fd.n, -1, np.int32)
        a = dict(px=_f32(mp["proj_x"]), py=_f32(mp["proj_y"]), pxr=_f32(mp.get("proj_xr")), lv=_i32(mp["level"]),
                 vc=_f32(mp["view_cos"]), d=_u8(mp["desc"]), iv=_u8(mp.get("in_view")), ho=_u8(mp.get("has_obs")))
        occ = _u8(frame_occupied)
        n = check(self._L.orbx_search_by_projection_mappoints(
            self._h, C.byref(fd), ptr(occ), n_mp, ptr(a["px"]), ptr(a["py"]), ptr(a["pxr"]), ptr(a["lv"]),
            ptr(a["vc"]), ptr(a["d"]), ptr(a["iv"]), ptr(a["ho"]), th, self.mfNNratio, ptr(fm)),
            "orbx_search_by_projection_mappoints")
        return n, fm

    # ---- SearchByProjection(Frame& Cur, const Frame& Last, th, bMono) (ORBmatcher.cc:1676-1887) ----
    def SearchByProjectionFrame(self, Cur: FrameView, q: dict, th: float, level_mode: int = 0, cur_occupied=None):
        """q: u, v, ur, octave, angle, desc, has_obs (last-frame map points already projected into Cur)."""
        fd = Cur.c_struct()
        nq = len(q["u"])
        cm = np.full(fd.n, -1, np.int32)
        a = dict(u=_f32(q["u"]), v=_f32(q["v"]), ur=_f32(q.get("ur")), o=_i32(q["octave"]), ang=_f32(q["angle"]),
                 d=_u8(q["desc"]), ho=_u8(q.get("has_obs")))
        occ = _u8(cur_occupied)
        n = check(self._L.orbx_search_by_projection_frame(
            self._h, C.byref(fd), ptr(occ), nq, ptr(a["u"]), ptr(a["v"]), ptr(a["ur"]), ptr(a["o"]), ptr(a["ang"]),
            ptr(a["d"]), ptr(a["ho"]), th, level_mode, int(self.mbCheckOrientation), ptr(cm)),
            "orbx_search_by_projection_frame")
        return n, cm
