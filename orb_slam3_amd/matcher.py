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
from ._lib import KP_DTYPE, FeatVec, FrameDesc, PAIR_PREDICATE, PinholeGate, check, ptr


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


class FeatureVector:
    """DBoW2::FeatureVector as arrays: `nodes` ascending node ids, `lists[k]` = feature indices of node k."""

    def __init__(self, nodes, lists):
        self.node_id = np.ascontiguousarray(nodes, np.uint32)
        assert np.all(np.diff(self.node_id.astype(np.int64)) > 0), "node ids must be strictly ascending (std::map order)"
        self.node_ptr = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int32)
        self.index = (np.concatenate(lists) if len(lists) else np.zeros(0)).astype(np.int32)

    @classmethod
    def from_node_of_feature(cls, node_of_feature):
        """Build from a per-feature node id array (features listed in ascending index inside a node, as
        TemplatedVocabulary::transform / FeatureVector::addFeature produce)."""
        node_of_feature = np.asarray(node_of_feature)
        nodes = np.unique(node_of_feature)
        return cls(nodes, [np.nonzero(node_of_feature == nd)[0] for nd in nodes])

    def c_struct(self, cls=FeatVec):
        return cls(self.node_id.ctypes.data, self.node_ptr.ctypes.data, self.index.ctypes.data, len(self.node_id))


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, np.float32)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, np.int32)


def _u8(a):
    return None if a is None else np.ascontiguousarray(a, np.uint8)


class ORBVocabulary:
    """DBoW2 vocabulary tree on the device (flattened: child CSR, node descriptors, leaf word ids)."""

    def __init__(self, L: int, child_ptr, child_idx, node_desc, word_id, device: int = 0):
        self._L_ = _lib.lib()
        self.L = L
        cp, ci = _i32(child_ptr), _i32(child_idx)
        nd, wi = _u8(node_desc), _i32(word_id)
        self._h = C.c_void_p()
        check(self._L_.orbx_vocabulary_create(device, L, len(wi), ptr(cp), ptr(ci), ptr(nd), ptr(wi), C.byref(self._h)),
              "orbx_vocabulary_create")

    def __del__(self):
        if getattr(self, "_h", None):
            self._L_.orbx_vocabulary_destroy(self._h)
            self._h = None


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

    def last_transfers(self) -> dict:
        """Transfers of the last call: orbx_matcher_debug_transfers (runs up / down, their bytes, how many went through a DMA engine, k_xfer launches)."""
        out = np.zeros(6, np.int64)
        n = self._L.orbx_matcher_debug_transfers(self._h, ptr(out), 6)
        if n < 0:
            raise RuntimeError(f"orbx_matcher_debug_transfers: {n}")
        return {"uploads": int(out[0]), "downloads": int(out[1]), "upload_bytes": int(out[2]), "download_bytes": int(out[3]),
                "dma_submissions": int(out[4]), "xfer_launches": int(out[5])}

    def last_replay_stats(self) -> dict:
        """k_replay_init_lists of the last SearchForInitialization: orbx_matcher_debug_replay_stats."""
        out = np.zeros(3, np.int32)
        if self._L.orbx_matcher_debug_replay_stats(self._h, ptr(out)) < 0:
            raise RuntimeError("orbx_matcher_debug_replay_stats")
        return {"rounds": int(out[0]), "rescans": int(out[1]), "queries": int(out[2])}

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

    def compute_stereo_fisheye_matches(self, kl, dl, mono_left, kr, dr, mono_right, level_sigma2, triangulate):
        """Frame::ComputeStereoFishEyeMatches (Frame.cc:1126-1166): kNN-2 of the lapping-area tails on the device (:1144), Lowe's ratio
        (:1151) and the bookkeeping (:1157-1162) on the host.  triangulate(i_left, i_right, sigma1, sigma2) -> (depth, (x, y, z)) is the
        caller's KannalaBrandt8::TriangulateMatches.  Returns (nMatches, descMatches, l2r, r2l, depth, u_right, p3d)."""
        kl, kr = np.ascontiguousarray(kl, KP_DTYPE), np.ascontiguousarray(kr, KP_DTYPE)
        dl, dr, s2 = _u8(dl), _u8(dr), _f32(level_sigma2)
        n_left, n_right = len(kl), len(kr)
        l2r, r2l = np.full(n_left, -1, np.int32), np.full(n_right, -1, np.int32)
        depth, ur, p3d = np.full(n_left, -1.0, np.float32), np.full(n_left, -1.0, np.float32), np.zeros((n_left, 3), np.float32)
        n_matches = n_desc = 0
        if n_left - mono_left > 0:
            idx, dist = self.knn2(dl[mono_left:], dr[mono_right:])
            for q in range(n_left - mono_left):
                if idx[q, 1] < 0 or not (float(np.float32(dist[q, 0])) < float(np.float32(dist[q, 1])) * 0.7):
                    continue
                n_desc += 1
                il, ir = q + mono_left, int(idx[q, 0]) + mono_right
                d, p = triangulate(il, ir, float(s2[kl["octave"][il]]), float(s2[kr["octave"][ir]]))
                d = np.float32(d)
                if d > np.float32(0.0001):
                    l2r[il], r2l[ir], depth[il], p3d[il] = ir, il, d, np.asarray(p, np.float32)
                    n_matches += 1
        return n_matches, n_desc, l2r, r2l, depth, ur, p3d

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
    def SearchByProjectionFrame(self, Cur: FrameView, q: dict, th: float, level_mode: int = 0, cur_occupied=None, raw=False):
        """q: u, v, ur, octave, angle, desc, has_obs (last-frame map points already projected into Cur).
        raw=True keeps the C ABI's -2 for "assigned, then cleared by the rotation check" (the slot becomes NULL in the reference)."""
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
        return n, (cm if raw else np.maximum(cm, -1))

    # ---- candidate generation: Frame::UndistortKeyPoints / ComputeImageBounds / isInFrustum ----
    def UndistortKeyPoints(self, cam, kps):
        from ._lib import Camera
        k = np.ascontiguousarray(kps, KP_DTYPE)
        out = np.zeros(len(k), KP_DTYPE)
        c = Camera(*[float(x) for x in cam])
        check(self._L.orbx_undistort_keypoints(self._h, C.byref(c), ptr(k), len(k), ptr(out)), "orbx_undistort_keypoints")
        return out

    def ComputeImageBounds(self, cam, width, height):
        from ._lib import Camera
        b = np.zeros(4, np.float32)
        c = Camera(*[float(x) for x in cam])
        check(self._L.orbx_image_bounds(C.byref(c), int(width), int(height), ptr(b)), "orbx_image_bounds")
        return b

    def isInFrustum(self, cam, pose, bounds, log_scale_factor, nlevels, cos_limit, pos, normal, min_dist, max_dist):
        """Frame::isInFrustum for n map points (Nleft == -1).  pose = (Rcw, tcw, Ow).  Returns dict like the oracle's."""
        from ._lib import Camera, FramePose
        P, Nn = _f32(np.asarray(pos).reshape(-1, 3)), _f32(np.asarray(normal).reshape(-1, 3))
        mn, mx, b = _f32(min_dist), _f32(max_dist), _f32(bounds)
        n = len(P)
        # (np.empty: the call writes every entry of every array)
        out = dict(in_view=np.empty(n, np.uint8), proj_x=np.empty(n, np.float32), proj_y=np.empty(n, np.float32), proj_xr=np.empty(n, np.float32),
                   depth=np.empty(n, np.float32), level=np.empty(n, np.int32), view_cos=np.empty(n, np.float32))
        c, p = Camera(*[float(x) for x in cam]), FramePose.make(*pose)
        check(self._L.orbx_is_in_frustum(self._h, C.byref(c), C.byref(p), ptr(b), float(log_scale_factor), int(nlevels), float(cos_limit), n, ptr(P),
                                         ptr(Nn), ptr(mn), ptr(mx), *[ptr(out[k]) for k in ("in_view", "proj_x", "proj_y", "proj_xr", "depth", "level", "view_cos")]),
              "orbx_is_in_frustum")
        return out

    def isInFrustumChecks(self, views, bounds, log_scale_factor, nlevels, cos_limit, pos, normal, min_dist, max_dist):
        """Frame::isInFrustumChecks (Frame.cc:1168) for n map points and the 1 or 2 cameras of a fisheye rig.  views = [(R, t, twc, params8), ...] as the
        reference's lines 1172-1186 compute them.  Returns dict of [n_views][n] arrays: in_view, proj_x, proj_y, depth, level, view_cos."""
        P, Nn = _f32(np.asarray(pos).reshape(-1, 3)), _f32(np.asarray(normal).reshape(-1, 3))
        mn, mx, b = _f32(min_dist), _f32(max_dist), _f32(bounds)
        n, nv = len(P), len(views)
        V = np.ascontiguousarray(np.concatenate([np.concatenate([np.asarray(x, np.float32).ravel() for x in v]) for v in views]), np.float32)
        assert V.size == 23 * nv
        out = dict(in_view=np.empty((nv, n), np.uint8), proj_x=np.empty((nv, n), np.float32), proj_y=np.empty((nv, n), np.float32),
                   depth=np.empty((nv, n), np.float32), level=np.empty((nv, n), np.int32), view_cos=np.empty((nv, n), np.float32))
        check(self._L.orbx_is_in_frustum_checks(self._h, ptr(V), nv, ptr(b), float(log_scale_factor), int(nlevels), float(cos_limit), n, ptr(P), ptr(Nn),
                                                ptr(mn), ptr(mx), *[ptr(out[k]) for k in ("in_view", "proj_x", "proj_y", "depth", "level", "view_cos")]),
              "orbx_is_in_frustum_checks")
        return out

    # ---- fisheye-stereo twins (F.Nleft != -1): features [0, n_left) left camera, [n_left, N) right camera ----
    def SearchByProjectionFisheye(self, left: FrameView, kps_right, l2r, r2l, mp: dict, th: float = 3.0, frame_occupied=None):
        """ORBmatcher.cc:43-213 whole.  left.descriptors holds ALL n_left + n_right rows; mp: in_view, proj_x, proj_y, level,
        view_cos, in_view_r, proj_xr, proj_yr, level_r, view_cos_r, desc, has_obs."""
        kr = np.ascontiguousarray(kps_right, KP_DTYPE)
        left.keypoints_un = np.ascontiguousarray(left.keypoints_un, KP_DTYPE)
        fd = left.c_struct()
        N = fd.n + len(kr)
        fm = np.full(N, -1, np.int32)
        a = [_u8(mp["in_view"]), _f32(mp["proj_x"]), _f32(mp["proj_y"]), _i32(mp["level"]), _f32(mp["view_cos"]), _u8(mp["in_view_r"]),
             _f32(mp["proj_xr"]), _f32(mp["proj_yr"]), _i32(mp["level_r"]), _f32(mp["view_cos_r"]), _u8(mp["desc"]), _u8(mp.get("has_obs"))]
        l2r, r2l, occ = _i32(l2r), _i32(r2l), _u8(frame_occupied)
        n = check(self._L.orbx_search_by_projection_mappoints_fisheye(self._h, C.byref(fd), ptr(kr), len(kr), ptr(l2r), ptr(r2l), ptr(occ), len(a[0]),
                                                                      *[ptr(x) for x in a], th, self.mfNNratio, ptr(fm)),
                  "orbx_search_by_projection_mappoints_fisheye")
        return n, fm

    def SearchByProjectionFrameFisheye(self, left: FrameView, kps_right, q: dict, th: float, level_mode: int = 0, cur_occupied=None, raw=False):
        """ORBmatcher.cc:1676-1887 with the twin :1794-1863.  q: u, v, xr, yr, octave, angle, desc, has_obs."""
        kr = np.ascontiguousarray(kps_right, KP_DTYPE)
        fd = left.c_struct()
        cm = np.full(fd.n + len(kr), -1, np.int32)
        a = [_f32(q["u"]), _f32(q["v"]), _f32(q["xr"]), _f32(q["yr"]), _i32(q["octave"]), _f32(q["angle"]), _u8(q["desc"]), _u8(q.get("has_obs"))]
        occ = _u8(cur_occupied)
        n = check(self._L.orbx_search_by_projection_frame_fisheye(self._h, C.byref(fd), ptr(kr), len(kr), ptr(occ), len(a[0]), *[ptr(x) for x in a],
                                                                  th, level_mode, int(self.mbCheckOrientation), ptr(cm)),
                  "orbx_search_by_projection_frame_fisheye")
        return n, (cm if raw else np.maximum(cm, -1))

    def SearchByBoWFrameFisheye(self, kf_desc, kf_angle, kf_valid, kf_fv: "FeatureVector", f_desc, f_angle, n_f_left: int, f_fv: "FeatureVector"):
        """ORBmatcher.cc:283-392 (frame features >= n_f_left are the right camera's; `|| true` on the right ratio test)."""
        kd, ka, kv, fdsc, fa = _u8(kf_desc), _f32(kf_angle), _u8(kf_valid), _u8(f_desc), _f32(f_angle)
        a, b = kf_fv.c_struct(), f_fv.c_struct()
        fm = np.full(len(fdsc), -1, np.int32)
        n = check(self._L.orbx_search_by_bow_frame_fisheye(self._h, ptr(kd), ptr(ka), ptr(kv), len(kd), C.byref(a), ptr(fdsc), ptr(fa), len(fdsc),
                                                           int(n_f_left), C.byref(b), self.mfNNratio, int(self.mbCheckOrientation), ptr(fm)),
                  "orbx_search_by_bow_frame_fisheye")
        return n, fm

    # ---- general window form: M3 = SearchByProjection(Frame&, KeyFrame*, ...) (ORBmatcher.cc:1889-2010) and
    #      M4 = SearchByProjection(KeyFrame*, Sim3f&, ...) (ORBmatcher.cc:427-646) ----
    def SearchByProjectionWindow(self, F: FrameView, q: dict, max_dist: float, check_orientation: bool, occupied=None, raw=False):
        """q: x, y, r, min_level, max_level, angle, desc[, has_obs].  raw: see SearchByProjectionFrame."""
        fd = F.c_struct()
        nq = len(q["x"])
        match = np.full(fd.n, -1, np.int32)
        a = dict(x=_f32(q["x"]), y=_f32(q["y"]), r=_f32(q["r"]), lo=_i32(q["min_level"]), hi=_i32(q["max_level"]),
                 ang=_f32(q.get("angle")), d=_u8(q["desc"]), ho=_u8(q.get("has_obs")))
        occ = _u8(occupied)
        n = check(self._L.orbx_search_by_projection_window(
            self._h, C.byref(fd), ptr(occ), nq, ptr(a["x"]), ptr(a["y"]), ptr(a["r"]), ptr(a["lo"]), ptr(a["hi"]),
            ptr(a["ang"]), ptr(a["d"]), ptr(a["ho"]), max_dist, int(check_orientation), ptr(match)),
            "orbx_search_by_projection_window")
        return n, (match if raw else np.maximum(match, -1))

    # ---- SearchForInitialization (ORBmatcher.cc:648-763) ----
    def SearchForInitialization(self, kps1_un, desc1, F2: FrameView, vbPrevMatched, windowSize=100):
        """Returns (nmatches, vnMatches12); vbPrevMatched (n1 x 2 float32) is updated in place."""
        k1 = np.ascontiguousarray(kps1_un, KP_DTYPE)
        d1 = _u8(desc1)
        assert vbPrevMatched.dtype == np.float32 and vbPrevMatched.flags.c_contiguous
        fd = F2.c_struct()
        m12 = np.full(len(k1), -1, np.int32)
        n = check(self._L.orbx_search_for_initialization(self._h, ptr(k1), ptr(d1), len(k1), C.byref(fd), ptr(vbPrevMatched),
                                                         int(windowSize), self.mfNNratio, int(self.mbCheckOrientation),
                                                         ptr(m12)), "orbx_search_for_initialization")
        return n, m12

    # ---- SearchByBoW (ORBmatcher.cc:223-425 / 765-905) ----
    def SearchByBoWFrame(self, kf_desc, kf_angle, kf_valid, kf_fv: FeatureVector, f_desc, f_angle, f_fv: FeatureVector):
        kd, ka, kv, fdsc, fa = _u8(kf_desc), _f32(kf_angle), _u8(kf_valid), _u8(f_desc), _f32(f_angle)
        a, b = kf_fv.c_struct(), f_fv.c_struct()
        fm = np.full(len(fdsc), -1, np.int32)
        n = check(self._L.orbx_search_by_bow_frame(self._h, ptr(kd), ptr(ka), ptr(kv), len(kd), C.byref(a), ptr(fdsc), ptr(fa),
                                                   len(fdsc), C.byref(b), self.mfNNratio, int(self.mbCheckOrientation),
                                                   ptr(fm)), "orbx_search_by_bow_frame")
        return n, fm

    def SearchByBoWKeyFrames(self, desc1, angle1, valid1, fv1: FeatureVector, desc2, angle2, valid2, fv2: FeatureVector):
        d1, a1, v1, d2, a2, v2 = _u8(desc1), _f32(angle1), _u8(valid1), _u8(desc2), _f32(angle2), _u8(valid2)
        a, b = fv1.c_struct(), fv2.c_struct()
        m12 = np.full(len(d1), -1, np.int32)
        n = check(self._L.orbx_search_by_bow_keyframes(self._h, ptr(d1), ptr(a1), ptr(v1), len(d1), C.byref(a), ptr(d2),
                                                       ptr(a2), ptr(v2), len(d2), C.byref(b), self.mfNNratio,
                                                       int(self.mbCheckOrientation), ptr(m12)),
                  "orbx_search_by_bow_keyframes")
        return n, m12

    # ---- SearchForTriangulation (ORBmatcher.cc:907-1146) ----
    def SearchForTriangulation(self, desc1, angle1, skip1, fv1: FeatureVector, desc2, angle2, skip2, fv2: FeatureVector,
                               pair_ok=None):
        """pair_ok(idx1, idx2) -> bool: the geometric gates (epipole distance + epipolarConstrain, or bCoarse)."""
        d1, a1, s1, d2, a2, s2 = _u8(desc1), _f32(angle1), _u8(skip1), _u8(desc2), _f32(angle2), _u8(skip2)
        a, b = fv1.c_struct(), fv2.c_struct()
        m12 = np.full(len(d1), -1, np.int32)
        cb = PAIR_PREDICATE((lambda user, i, j: int(bool(pair_ok(i, j)))) if pair_ok else 0)
        n = check(self._L.orbx_search_for_triangulation(self._h, ptr(d1), ptr(a1), ptr(s1), len(d1), C.byref(a), ptr(d2),
                                                        ptr(a2), ptr(s2), len(d2), C.byref(b), int(self.mbCheckOrientation),
                                                        cb, None, ptr(m12)), "orbx_search_for_triangulation")
        return n, m12

    def SearchForTriangulationPinhole(self, kps1_un, desc1, skip1, fv1: FeatureVector, kps2_un, desc2, skip2, fv2: FeatureVector,
                                      scale_factors2, level_sigma2_2, F12, epipole, u_right1=None, u_right2=None, coarse=False,
                                      strict_fp=False):
        """SearchForTriangulation for pinhole key frames with the epipole-distance test (ORBmatcher.cc:1026-1034) and
        Pinhole::epipolarConstrain (Pinhole.cpp:107-129, on the caller's F12) evaluated on the device: no callback."""
        k1, k2 = np.ascontiguousarray(kps1_un, KP_DTYPE), np.ascontiguousarray(kps2_un, KP_DTYPE)
        d1, s1, d2, s2 = _u8(desc1), _u8(skip1), _u8(desc2), _u8(skip2)
        sf, sg = _f32(scale_factors2), _f32(level_sigma2_2)
        ur1, ur2 = _f32(u_right1), _f32(u_right2)
        g = PinholeGate(k1.ctypes.data, k2.ctypes.data, None if ur1 is None else ur1.ctypes.data, None if ur2 is None else ur2.ctypes.data,
                        sf.ctypes.data, sg.ctypes.data, len(sf), (C.c_float * 9)(*[float(x) for x in np.asarray(F12, np.float32).ravel()]),
                        float(epipole[0]), float(epipole[1]), int(coarse), int(strict_fp))
        a, b = fv1.c_struct(), fv2.c_struct()
        m12 = np.full(len(k1), -1, np.int32)
        n = check(self._L.orbx_search_for_triangulation_pinhole(self._h, ptr(d1), ptr(s1), len(k1), C.byref(a), ptr(d2), ptr(s2), len(k2),
                                                                C.byref(b), int(self.mbCheckOrientation), C.byref(g), ptr(m12)),
                  "orbx_search_for_triangulation_pinhole")
        return n, m12

    def SearchForTriangulationKB8(self, kps1, n_left1, desc1, skip1, fv1: FeatureVector, kps2, n_left2, desc2, skip2, fv2: FeatureVector,
                                  level_sigma2_1, level_sigma2_2, cam1, cam2, R12, t12, coarse=False):
        """SearchForTriangulation between key frames of a fisheye rig: KannalaBrandt8::epipolarConstrain (ORBmatcher.cc:1036-1072) on the device.
        kps: mvKeys | mvKeysRight; cam1 / cam2: [2][8] parameters of (mpCamera, mpCamera2); R12 [4][3][3], t12 [4][3] = ll, lr, rl, rr."""
        from ._lib import Kb8GateStruct
        k1, k2 = np.ascontiguousarray(kps1, KP_DTYPE), np.ascontiguousarray(kps2, KP_DTYPE)
        d1, s1, d2, s2 = _u8(desc1), _u8(skip1), _u8(desc2), _u8(skip2)
        sg1, sg2 = _f32(level_sigma2_1), _f32(level_sigma2_2)
        flat = lambda x, n: (C.c_float * n)(*[float(v) for v in np.asarray(x, np.float32).ravel()])
        g = Kb8GateStruct(k1.ctypes.data, k2.ctypes.data, int(n_left1), int(n_left2), sg1.ctypes.data, sg2.ctypes.data, len(sg1), flat(cam1, 16), flat(cam2, 16),
                          flat(R12, 36), flat(t12, 12), int(coarse))
        a, b = fv1.c_struct(), fv2.c_struct()
        m12 = np.full(len(k1), -1, np.int32)
        n = check(self._L.orbx_search_for_triangulation_kb8(self._h, ptr(d1), ptr(s1), len(k1), C.byref(a), ptr(d2), ptr(s2), len(k2), C.byref(b),
                                                            int(self.mbCheckOrientation), C.byref(g), ptr(m12)), "orbx_search_for_triangulation_kb8")
        return n, m12

    def DebugKb8Epipolar(self, cam1, cam2, R12, t12, xy1, xy2, sigma1, sigma2, sel):
        """Test hook: KannalaBrandt8::epipolarConstrain of n independent pairs on the device (orbx_debug_kb8_epipolar); returns ok uint8[n]."""
        c1, c2, R, t = (np.ascontiguousarray(x, np.float32).ravel() for x in (cam1, cam2, R12, t12))
        assert len(c1) == 16 and len(c2) == 16 and len(R) == 36 and len(t) == 12
        a, b = np.ascontiguousarray(xy1, np.float32).reshape(-1, 2), np.ascontiguousarray(xy2, np.float32).reshape(-1, 2)
        s1, s2, se = _f32(sigma1), _f32(sigma2), _u8(sel)
        ok = np.zeros(len(a), np.uint8)
        check(self._L.orbx_debug_kb8_epipolar(self._h, ptr(c1), ptr(c2), ptr(R), ptr(t), len(a), ptr(a), ptr(b), ptr(s1), ptr(s2), ptr(se), ptr(ok)),
              "orbx_debug_kb8_epipolar")
        return ok

    # ---- DBoW2 transform (Frame::ComputeBoW, Frame.cc:738-745) ----
    def BowTransform(self, voc: "ORBVocabulary", descriptors, levelsup: int = 4):
        """Returns (word_id[n], node_id[n]) of TemplatedVocabulary::transform for every descriptor."""
        d = _u8(descriptors)
        w = np.zeros(len(d), np.int32)
        nd = np.zeros(len(d), np.int32)
        check(self._L.orbx_bow_transform(self._h, voc._h, ptr(d), len(d), levelsup, ptr(w), ptr(nd)), "orbx_bow_transform")
        return w, nd

    # ---- MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:329-403), batched ----
    def DistinctiveDescriptors(self, descriptors, set_ptr):
        d, sp = _u8(descriptors), _i32(set_ptr)
        out = np.zeros(len(sp) - 1, np.int32)
        check(self._L.orbx_distinctive_descriptors(self._h, ptr(d), ptr(sp), len(sp) - 1, ptr(out)), "orbx_distinctive_descriptors")
        return out

    # ---- matching core of Fuse x2 (ORBmatcher.cc:1148-1455) ----
    def FuseSearch(self, KF: FrameView, q: dict, inv_level_sigma2=None, strict_fp: bool = False):
        """q: u, v, ur, r, level, desc.  Returns (best_idx[nq], best_dist[nq])."""
        fd = KF.c_struct()
        nq = len(q["u"])
        a = dict(u=_f32(q["u"]), v=_f32(q["v"]), ur=_f32(q.get("ur")), r=_f32(q["r"]), lv=_i32(q["level"]), d=_u8(q["desc"]))
        isg = _f32(inv_level_sigma2)
        bi = np.zeros(nq, np.int32)
        bd = np.zeros(nq, np.int32)
        check(self._L.orbx_fuse_search(self._h, C.byref(fd), ptr(isg), nq, ptr(a["u"]), ptr(a["v"]), ptr(a["ur"]), ptr(a["r"]),
                                       ptr(a["lv"]), ptr(a["d"]), int(strict_fp), ptr(bi), ptr(bd)), "orbx_fuse_search")
        return bi, bd

    # ---- SearchBySim3 (ORBmatcher.cc:1457-1674): two gate-less fuse searches + mutual agreement ----
    def SearchBySim3(self, KF1: FrameView, KF2: FrameView, side1: dict, side2: dict, th: float, already_matched1=None,
                     already_matched2=None, scale_factors1=None, scale_factors2=None):
        """side1: for every feature of KF1, its map point transformed by S21 and projected into KF2 — dict(valid, u, v, level,
        desc): valid[i] = the slot holds a good map point that survived the depth / image / distance gates of :1500-1527,
        (u, v) the projection, level = PredictScale(dist3D, pKF2), desc = pMP->GetDescriptor().  side2: the same for KF2's
        map points projected into KF1 (:1576-1645).  already_matched1/2 = vbAlreadyMatched1/2 (:1477-1490).  Search radius
        th * mvScaleFactors[level] of the key frame searched in.  Returns (nFound, match12[n1]) with match12[i1] = KF2 feature
        index (the reference stores vpMapPoints2[idx2]) or -1; slots the caller had matched already are left at -1."""
        sf1 = _f32(KF1.scale_factors if scale_factors1 is None else scale_factors1)
        sf2 = _f32(KF2.scale_factors if scale_factors2 is None else scale_factors2)

        def one_way(side, target: FrameView, sf, done):
            lv = _i32(side["level"])
            ok = _u8(side["valid"]) == 1
            if done is not None:
                ok &= _u8(done) == 0
            q = dict(u=side["u"], v=side["v"], ur=np.zeros(len(lv), np.float32), r=(np.float32(th) * sf[lv]).astype(np.float32),
                     level=lv, desc=side["desc"])
            bi, bd = self.FuseSearch(target, q, None)
            return np.where(ok & (bd <= 100), bi, -1)          # bestDist <= TH_HIGH (:1569, :1656)
        m1 = one_way(side1, KF2, sf2, already_matched1)
        m2 = one_way(side2, KF1, sf1, already_matched2)
        back = np.where(m1 >= 0, m2[np.maximum(m1, 0)], -2)
        match12 = np.where(back == np.arange(len(m1)), m1, -1).astype(np.int32)   # :1662-1675 agreement check
        return int((match12 >= 0).sum()), match12
