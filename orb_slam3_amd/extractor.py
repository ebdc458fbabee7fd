"""Host-side mirror of ORB_SLAM3::ORBextractor (/root/reference/include/ORBextractor.h:43-109) over the C ABI.

Same constructor arguments, same accessor names, same call result as the reference class; the work is done
by the HIP kernels in liborbx.so (no CPU path).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, Params, check, ptr


class ORBextractor:
    HARRIS_SCORE = 0
    FAST_SCORE = 1

    def __init__(self, nfeatures: int, scaleFactor: float, nlevels: int, iniThFAST: int, minThFAST: int,
                 device: int = 0, flags: int = 0):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        prm = Params(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, flags)
        check(self._L.orbx_create(C.byref(prm), device, 0, 0, 0, C.byref(self._h)), "orbx_create")
        self.nfeatures, self.nlevels, self.device = nfeatures, nlevels, device
        self._last_shape = None
        self._cap_of = {}   # (width, height) -> output capacity

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orbx_destroy(self._h)
            self._h = None

    # ---- ORBextractor::operator() (ORBextractor.cc:1086-1168) ----
    def __call__(self, image: np.ndarray, mask=None, vLappingArea=(0, 0)):
        """Returns (monoIndex, keypoints[KP_DTYPE], descriptors[N,32] uint8); monoIndex == -1 for an empty image."""
        if image is None or image.size == 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 expected (ORBextractor.cc:1094)"
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        cap = self._cap_of.get((w, h))
        if cap is None:
            cap = self._cap_of[(w, h)] = check(self._L.orbx_output_capacity(self._h, w, h), "orbx_output_capacity")
        kps = np.empty(cap, KP_DTYPE)
        desc = np.empty((cap, 32), np.uint8)
        n, mono = C.c_int(0), C.c_int(0)
        st = self._L.orbx_extract(self._h, ptr(image), w, h, image.strides[0], int(vLappingArea[0]),
                                  int(vLappingArea[1]), ptr(kps), ptr(desc), cap, C.byref(n), C.byref(mono))
        if st == _lib.ORBX_E_EMPTY:
            return -1, kps[:0], desc[:0]
        check(st, "orbx_extract")
        self._last_shape = (w, h)
        return mono.value, kps[:n.value].copy(), desc[:n.value].copy()

    # ---- batched, device-resident form ----
    def extract_batch_device(self, d_ptr: int, n_frames: int, width: int, height: int, row_stride: int,
                             frame_stride: int, vLappingArea=(0, 0)):
        check(self._L.orbx_extract_batch_device(self._h, C.c_void_p(d_ptr), n_frames, width, height, row_stride,
                                                frame_stride, int(vLappingArea[0]), int(vLappingArea[1])),
              "orbx_extract_batch_device")
        self._last_shape = (width, height)

    def extract_batch_host(self, h_ptr: int, n_frames: int, width: int, height: int, row_stride: int, frame_stride: int,
                           vLappingArea=(0, 0)):
        """Frames in PINNED host memory (hipHostMalloc / hipHostRegister, e.g. torch's pin_memory(); pageable memory is refused with
        ORBX_E_BAD_ARG); the upload overlaps the previous batch's kernels.  The frames may be overwritten once the NEXT call has returned."""
        check(self._L.orbx_extract_batch_host(self._h, C.c_void_p(h_ptr), n_frames, width, height, row_stride,
                                              frame_stride, int(vLappingArea[0]), int(vLappingArea[1])),
              "orbx_extract_batch_host")
        self._last_shape = (width, height)

    def sync(self):
        check(self._L.orbx_sync(self._h), "orbx_sync")

    def output_capacity(self, width, height) -> int:
        return check(self._L.orbx_output_capacity(self._h, width, height), "orbx_output_capacity")

    def batch_view(self):
        v = _lib.BatchView()
        check(self._L.orbx_batch_view_get(self._h, C.byref(v)), "orbx_batch_view_get")
        return v

    def download(self, frame: int):
        cap = self.batch_view().cap
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n, mono = C.c_int(0), C.c_int(0)
        check(self._L.orbx_batch_download(self._h, frame, ptr(kps), ptr(desc), cap, C.byref(n), C.byref(mono)),
              "orbx_batch_download")
        return mono.value, kps[:n.value].copy(), desc[:n.value].copy()

    # ---- camera of the batch path: Frame::UndistortKeyPoints + ComputeImageBounds for the batched matchers ----
    def set_camera(self, cam=None):
        """cam: tuple (fx, fy, cx, cy, k1, k2, p1, p2, k3, bf) in that order, a dict with those keys, or None (distortion-free default)."""
        if cam is None:
            check(self._L.orbx_set_camera(self._h, None), "orbx_set_camera")
        else:
            if isinstance(cam, dict):
                cam = [cam[k] for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "bf")]
            c = _lib.Camera(*[float(x) for x in cam])
            check(self._L.orbx_set_camera(self._h, C.byref(c)), "orbx_set_camera")

    def download_keypoints_un(self, frame: int) -> np.ndarray:
        cap = self.batch_view().cap
        k = np.zeros(cap, KP_DTYPE)
        n = C.c_int(0)
        check(self._L.orbx_batch_download_keypoints_un(self._h, frame, ptr(k), cap, C.byref(n)), "orbx_batch_download_keypoints_un")
        return k[:n.value].copy()

    def frustum_batch_device(self, cam, poses, cos_limit, n_mp, d_pos, d_normal, d_min, d_max, d_in_view, d_proj_x, d_proj_y, d_proj_xr,
                             d_depth, d_level, d_view_cos, bounds=None):
        """Frame::isInFrustum for every (pose, map point); poses: list of (Rcw, tcw, Ow); d_*: raw device pointers."""
        c = _lib.Camera(*[float(x) for x in cam])
        P = (_lib.FramePose * len(poses))(*[_lib.FramePose.make(*p) for p in poses])
        b = None if bounds is None else np.ascontiguousarray(bounds, np.float32)
        vp = C.c_void_p
        check(self._L.orbx_frustum_batch_device(self._h, C.byref(c), P, len(poses), ptr(b), cos_limit, n_mp,
                                                *[vp(x) for x in (d_pos, d_normal, d_min, d_max, d_in_view, d_proj_x, d_proj_y, d_proj_xr, d_depth,
                                                                  d_level, d_view_cos)]), "orbx_frustum_batch_device")

    def download_all(self, kps=None, desc=None, counts=None, mono=None):
        v = self.batch_view()
        n, cap = v.n_frames, v.cap
        kps = np.zeros((n, cap), KP_DTYPE) if kps is None else kps
        desc = np.zeros((n, cap, 32), np.uint8) if desc is None else desc
        counts = np.zeros(n, np.int32) if counts is None else counts
        mono = np.zeros(n, np.int32) if mono is None else mono
        check(self._L.orbx_batch_download_all(self._h, ptr(kps), ptr(desc), ptr(counts), ptr(mono)),
              "orbx_batch_download_all")
        return kps, desc, counts, mono

    def download_async(self, kps, desc, counts, mono, match=None, nmatches=None):
        """Enqueue the D2H of the last batch on the copy stream (arguments: raw host pointers to PINNED memory; pageable memory is refused)."""
        check(self._L.orbx_batch_download_async(self._h, *[C.c_void_p(p) if p else None
                                                           for p in (kps, desc, counts, mono, match, nmatches)]),
              "orbx_batch_download_async")

    def download_wait(self):
        check(self._L.orbx_download_wait(self._h), "orbx_download_wait")

    def match_consecutive_device(self, d_match: int = 0, d_nmatches: int = 0, th=15.0, du=0.0, dv=0.0, check_orientation=True):
        check(self._L.orbx_match_consecutive_device(self._h, th, du, dv, int(check_orientation),
                                                    C.c_void_p(d_match) if d_match else None,
                                                    C.c_void_p(d_nmatches) if d_nmatches else None),
              "orbx_match_consecutive_device")

    def search_mappoints_batch_device(self, n_mp: int, d_proj_x: int, d_proj_y: int, d_level: int, d_view_cos: int, d_in_view: int,
                                      d_mp_desc: int, desc_frame_stride=None, th=1.0, nnratio=0.8, d_match: int = 0, d_nmatches: int = 0):
        """SearchByProjection(Frame, MapPoints) for every frame of the resident batch (device arrays [n_frames][n_mp])."""
        stride = 32 * n_mp if desc_frame_stride is None else desc_frame_stride
        vp = C.c_void_p
        check(self._L.orbx_search_mappoints_batch_device(self._h, n_mp, vp(d_proj_x), vp(d_proj_y), vp(d_level), vp(d_view_cos),
                                                         vp(d_in_view) if d_in_view else None, vp(d_mp_desc), stride, th, nnratio,
                                                         vp(d_match) if d_match else None, vp(d_nmatches) if d_nmatches else None),
              "orbx_search_mappoints_batch_device")

    def stereo_download_all(self, h_ur: int, h_depth: int, h_nm: int):
        check(self._L.orbx_stereo_batch_download_all(self._h, C.c_void_p(h_ur), C.c_void_p(h_depth), C.c_void_p(h_nm)),
              "orbx_stereo_batch_download_all")

    # ---- Frame::ComputeStereoMatches on two resident batches (self = left extractor) ----
    def stereo_download_async(self, h_ur: int, h_depth: int, h_nm: int):
        """mvuRight / mvDepth [n_frames][cap] and the match counts into PINNED host buffers behind the stereo kernels; stereo_download_wait()
        returns when they are complete.  The next pair of batches may be extracted meanwhile."""
        check(self._L.orbx_stereo_batch_download_async(self._h, C.c_void_p(h_ur), C.c_void_p(h_depth), C.c_void_p(h_nm)),
              "orbx_stereo_batch_download_async")

    def stereo_download_wait(self):
        check(self._L.orbx_stereo_download_wait(self._h), "orbx_stereo_download_wait")

    def stereo_batch_device(self, right: "ORBextractor", bf: float, b: float):
        check(self._L.orbx_stereo_batch_device(self._h, right._h, bf, b), "orbx_stereo_batch_device")

    def stereo_download(self, frame: int):
        """Returns (n_matches, mvuRight[N], mvDepth[N]) of `frame` (-1 where unmatched)."""
        cap = self.batch_view().cap
        ur = np.zeros(cap, np.float32)
        depth = np.zeros(cap, np.float32)
        nl, nm = C.c_int(0), C.c_int(0)
        check(self._L.orbx_stereo_batch_download(self._h, frame, ptr(ur), ptr(depth), C.byref(nl), C.byref(nm)),
              "orbx_stereo_batch_download")
        return nm.value, ur[:nl.value].copy(), depth[:nl.value].copy()

    # ---- accessors (ORBextractor.h:62-83) ----
    def GetLevels(self) -> int:
        return self._L.orbx_get_levels(self._h)

    def GetScaleFactor(self) -> float:
        return self._L.orbx_get_scale_factor(self._h)

    def _tables(self):
        n = self.nlevels
        t = [np.zeros(n, np.float32) for _ in range(4)]
        self._L.orbx_get_scale_tables(self._h, *[ptr(a) for a in t])
        return t

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def feature_tables(self):
        quota = np.zeros(self.nlevels, np.int32)
        umax = np.zeros(16, np.int32)
        self._L.orbx_get_feature_tables(self._h, ptr(quota), ptr(umax))
        return quota, umax

    def level_size(self, level: int, shape=None):
        w, h = shape or self._last_shape
        lw, lh = C.c_int(0), C.c_int(0)
        check(self._L.orbx_level_size(self._h, w, h, level, C.byref(lw), C.byref(lh)), "orbx_level_size")
        return lw.value, lh.value

    def get_level(self, level: int, frame: int = 0) -> np.ndarray:
        """Padded level (19-px ring) of `frame`: what mvImagePyramid[level]'s parent buffer holds."""
        w, h = self.level_size(level)
        out = np.zeros((h + 38, w + 38), np.uint8)
        check(self._L.orbx_get_level(self._h, frame, level, ptr(out), out.strides[0]), "orbx_get_level")
        return out

    @property
    def mvImagePyramid(self):
        """ROI views of the last single-frame pyramid (public member of the reference, ORBextractor.h:83)."""
        return [self.get_level(l)[19:-19, 19:-19] for l in range(self.nlevels)]

    # ---- stage introspection for parity tests ----
    def debug_candidates(self, level: int, frame: int = 0) -> np.ndarray:
        n = check(self._L.orbx_debug_level_candidates(self._h, frame, level, None, 0), "candidates")
        out = np.zeros(max(n, 1), KP_DTYPE)
        self._L.orbx_debug_level_candidates(self._h, frame, level, ptr(out), n)
        return out[:n]

    def debug_level_keypoints(self, level: int, frame: int = 0) -> np.ndarray:
        n = check(self._L.orbx_debug_level_keypoints(self._h, frame, level, None, 0), "level keypoints")
        out = np.zeros(max(n, 1), KP_DTYPE)
        self._L.orbx_debug_level_keypoints(self._h, frame, level, ptr(out), n)
        return out[:n]

    def debug_blurred(self, level: int, frame: int = 0) -> np.ndarray:
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        check(self._L.orbx_debug_level_blurred(self._h, frame, level, ptr(out), out.strides[0]), "blurred")
        return out

    def debug_fused_patches(self, frame: int = 0, cap: int = 4096) -> np.ndarray:
        """(n, 37, 37) blurred pixels around the keypoints of `frame` as k_describe_fused computed them in LDS (output order)."""
        out = np.zeros((cap, 37, 37), np.uint8)
        n = self._L.orbx_debug_fused_patches(self._h, frame, ptr(out), cap)
        if n < 0:
            raise RuntimeError(f"orbx_debug_fused_patches: {n}")
        return out[:n]

    def stage_stats(self) -> dict:
        """Which paths the last batch took: orbx_debug_stage_stats."""
        out = (C.c_int64 * 8)()
        n = self._L.orbx_debug_stage_stats(self._h, out, 8)
        if n < 7:
            raise RuntimeError(f"orbx_debug_stage_stats: {n}")
        return {"fast_list_cells": int(out[0]), "cells": int(out[1]), "quadtrees_le_1792": int(out[2]), "quadtrees_le_4096": int(out[3]),
                "quadtrees_single_wave": int(out[4]), "fast_candidates": int(out[5]), "max_candidates_of_a_level": int(out[6])}

    def tune_fast_queues(self, mode: int = 1) -> dict:
        """orbx_tune_fast_queues on the last batch: mode 0 report, 1 adapt (grow when > 10 % of the cells took the FAST list pass, shrink back when
        < 0.5 %), 2 defaults.  Results never depend on the queue size; dense texture is ~2 x faster with the queues it ends up with."""
        info = (C.c_int32 * 4)()
        r = self._L.orbx_tune_fast_queues(self._h, int(mode), info)
        if r < 0:
            raise RuntimeError(f"orbx_tune_fast_queues: {r}")
        return {"changed": bool(r), "fast_list_cells": int(info[0]), "cells": int(info[1]), "group_queue": int(info[2]), "pixel_queue": int(info[3])}

    def profile_enable(self, on=True):
        self._L.orbx_profile_enable(self._h, int(on))

    def profile_read(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_double * 16)()
        cnt = (C.c_int64 * 16)()
        k = self._L.orbx_profile_read(self._h, names, ms, cnt, 16)
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(k)}
