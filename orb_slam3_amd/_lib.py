"""ctypes loader for liborbx.so (the C ABI declared in include/orbx.h).

torch is imported first on purpose: PyTorch-ROCm bundles its own libamdhip64.so (same SONAME as /opt/rocm's);
loading liborbx after torch makes both share ONE HIP runtime in the process (the tests and bench.py hold device
buffers as torch tensors next to our own HIP streams; the cross-rank barrier of bench.py is a gloo host group,
no RCCL communicator -- orb_slam3_amd/sharding.py).  There is no CPU fallback:
a missing or unloadable library raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "liborbx.so"

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])

ORBX_OK = 0
ORBX_E_EMPTY = -1
ORBX_E_STALE = -9
FLAG_DESC_STRICT = 1
FLAG_BLUR_OCV440 = 2
FLAG_ATAN_FMA = 4
TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30


class OrbxError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        L = lib()
        msg = L.orbx_status_string(status).decode()
        detail = L.orbx_last_error().decode()
        super().__init__(f"{where}: {msg} ({status}) {detail}")


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("flags", C.c_uint32)]


class BatchView(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("cap", C.c_int32), ("d_keypoints", C.c_void_p),
                ("d_descriptors", C.c_void_p), ("d_count", C.c_void_p), ("d_mono_index", C.c_void_p),
                ("d_keypoints_un", C.c_void_p)]


class Camera(C.Structure):
    """orbx_camera: Pinhole intrinsics, radial-tangential distortion (k1, k2, p1, p2, k3), mbf."""
    _fields_ = [(k, C.c_float) for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "bf")]


class FramePose(C.Structure):
    """orbx_frame_pose: Frame::mRcw (row-major), mtcw, mOw."""
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3)]

    @classmethod
    def make(cls, Rcw, tcw, Ow):
        import numpy as np
        p = cls()
        p.Rcw[:] = [float(x) for x in np.asarray(Rcw, np.float32).ravel()]
        p.tcw[:] = [float(x) for x in np.asarray(tcw, np.float32).ravel()]
        p.Ow[:] = [float(x) for x in np.asarray(Ow, np.float32).ravel()]
        return p


class FrameDesc(C.Structure):
    _fields_ = [("keypoints_un", C.c_void_p), ("descriptors", C.c_void_p), ("n", C.c_int32),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("scale_factors", C.c_void_p), ("nlevels", C.c_int32), ("u_right", C.c_void_p)]


class PinholeGate(C.Structure):
    """orbx_pinhole_gate (include/orbx.h)."""
    _fields_ = [("kps1_un", C.c_void_p), ("kps2_un", C.c_void_p), ("u_right1", C.c_void_p), ("u_right2", C.c_void_p),
                ("scale_factors2", C.c_void_p), ("level_sigma2_2", C.c_void_p), ("nlevels", C.c_int), ("F12", C.c_float * 9),
                ("ep_x", C.c_float), ("ep_y", C.c_float), ("coarse", C.c_int), ("strict_fp", C.c_int)]


class Kb8GateStruct(C.Structure):
    """orbx_kb8_gate (include/orbx.h)."""
    _fields_ = [("kps1", C.c_void_p), ("kps2", C.c_void_p), ("n_left1", C.c_int), ("n_left2", C.c_int), ("level_sigma2_1", C.c_void_p),
                ("level_sigma2_2", C.c_void_p), ("nlevels", C.c_int), ("cam1", C.c_float * 16), ("cam2", C.c_float * 16), ("R12", C.c_float * 36),
                ("t12", C.c_float * 12), ("coarse", C.c_int)]


class FeatVec(C.Structure):
    """DBoW2::FeatureVector flattened: node ids ascending + CSR of feature indices."""
    _fields_ = [("node_id", C.c_void_p), ("node_ptr", C.c_void_p), ("index", C.c_void_p), ("n_nodes", C.c_int32)]


PAIR_PREDICATE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)

_lib = None

# every symbol include/orbx.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "orbx_create", "orbx_destroy", "orbx_extract", "orbx_extract_batch_device", "orbx_extract_batch_host", "orbx_batch_view_get", "orbx_sync",
    "orbx_batch_download", "orbx_batch_download_all", "orbx_batch_download_async", "orbx_download_wait", "orbx_output_capacity", "orbx_get_level", "orbx_level_size",
    "orbx_get_level_device", "orbx_get_levels", "orbx_get_scale_factor", "orbx_get_scale_tables",
    "orbx_get_feature_tables", "orbx_debug_level_candidates", "orbx_debug_level_keypoints",
    "orbx_debug_level_blurred", "orbx_debug_stage_stats", "orbx_tune_fast_queues", "orbx_debug_fused_patches", "orbx_profile_enable", "orbx_profile_read", "orbx_matcher_create",
    "orbx_matcher_destroy", "orbx_matcher_debug_transfers", "orbx_matcher_debug_replay_stats", "orbx_hamming_csr", "orbx_hamming_best2_csr", "orbx_knn2", "orbx_stereo_rowband",
    "orbx_compute_stereo_matches", "orbx_search_by_projection_mappoints", "orbx_search_by_projection_frame",
    "orbx_match_consecutive_device", "orbx_last_error", "orbx_status_string", "orbx_search_by_projection_window", "orbx_search_by_projection_mappoints_fisheye", "orbx_search_by_projection_frame_fisheye",
    "orbx_search_by_bow_frame_fisheye", "orbx_undistort_keypoints", "orbx_image_bounds", "orbx_is_in_frustum", "orbx_is_in_frustum_checks", "orbx_frustum_batch_device",
    "orbx_set_camera", "orbx_batch_download_keypoints_un",
    "orbx_search_for_initialization", "orbx_search_by_bow_frame", "orbx_search_by_bow_keyframes",
    "orbx_search_for_triangulation", "orbx_search_for_triangulation_pinhole", "orbx_search_for_triangulation_kb8", "orbx_debug_kb8_epipolar", "orbx_stereo_batch_device", "orbx_stereo_batch_download", "orbx_stereo_batch_download_all", "orbx_stereo_batch_download_async", "orbx_stereo_download_wait", "orbx_search_mappoints_batch_device", "orbx_vocabulary_create",
    "orbx_vocabulary_destroy", "orbx_bow_transform", "orbx_distinctive_descriptors", "orbx_fuse_search",
]


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  liborbx has no CPU fallback.")
    import torch  # noqa: F401  (see module docstring: share one HIP runtime with torch)
    L = C.CDLL(str(LIB_PATH))
    vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.orbx_last_error.restype = C.c_char_p
    L.orbx_status_string.restype = C.c_char_p
    L.orbx_status_string.argtypes = [i32]
    L.orbx_create.argtypes = [C.POINTER(Params), i32, i32, i32, i32, C.POINTER(vp)]
    L.orbx_destroy.argtypes = [vp]
    L.orbx_extract.argtypes = [vp, vp, i32, i32, sz, i32, i32, vp, vp, i32, vp, vp]
    L.orbx_extract_batch_device.argtypes = [vp, vp, i32, i32, i32, sz, sz, i32, i32]
    L.orbx_extract_batch_host.argtypes = [vp, vp, i32, i32, i32, sz, sz, i32, i32]
    L.orbx_batch_view_get.argtypes = [vp, C.POINTER(BatchView)]
    L.orbx_sync.argtypes = [vp]
    L.orbx_batch_download.argtypes = [vp, i32, vp, vp, i32, vp, vp]
    L.orbx_batch_download_all.argtypes = [vp, vp, vp, vp, vp]
    L.orbx_output_capacity.argtypes = [vp, i32, i32]
    L.orbx_batch_download_async.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.orbx_download_wait.argtypes = [vp]
    L.orbx_get_level.argtypes = [vp, i32, i32, vp, sz]
    L.orbx_level_size.argtypes = [vp, i32, i32, i32, vp, vp]
    L.orbx_get_level_device.argtypes = [vp, i32, i32, vp, vp]
    L.orbx_get_levels.argtypes = [vp]
    L.orbx_get_scale_factor.argtypes = [vp]
    L.orbx_get_scale_factor.restype = f32
    L.orbx_get_scale_tables.argtypes = [vp, vp, vp, vp, vp]
    L.orbx_get_feature_tables.argtypes = [vp, vp, vp]
    L.orbx_debug_level_candidates.argtypes = [vp, i32, i32, vp, i32]
    L.orbx_debug_level_keypoints.argtypes = [vp, i32, i32, vp, i32]
    L.orbx_debug_level_blurred.argtypes = [vp, i32, i32, vp, sz]
    L.orbx_debug_stage_stats.argtypes = [vp, vp, i32]
    L.orbx_tune_fast_queues.argtypes = [vp, i32, vp]
    L.orbx_debug_fused_patches.argtypes = [vp, i32, vp, i32]
    L.orbx_debug_sort_nodes.argtypes = [i32, vp, vp, i32, vp]
    L.orbx_debug_sort_nodes_par.argtypes = [i32, vp, vp, i32, vp, vp]
    L.orbx_profile_enable.argtypes = [vp, i32]
    L.orbx_profile_read.argtypes = [vp, vp, vp, vp, i32]
    L.orbx_matcher_create.argtypes = [i32, C.POINTER(vp)]
    L.orbx_matcher_destroy.argtypes = [vp]
    L.orbx_matcher_debug_transfers.argtypes = [vp, vp, i32]
    L.orbx_matcher_debug_replay_stats.argtypes = [vp, vp]
    L.orbx_hamming_csr.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp]
    L.orbx_hamming_best2_csr.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp]
    L.orbx_knn2.argtypes = [vp, vp, i32, vp, i32, vp, vp]
    L.orbx_stereo_rowband.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, i32, i32, f32, f32, vp, vp]
    L.orbx_compute_stereo_matches.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, f32, f32,
                                              vp, vp]
    L.orbx_search_by_projection_mappoints.argtypes = [vp, C.POINTER(FrameDesc), vp, i32, vp, vp, vp, vp, vp, vp, vp,
                                                      vp, f32, f32, vp]
    L.orbx_search_by_projection_frame.argtypes = [vp, C.POINTER(FrameDesc), vp, i32, vp, vp, vp, vp, vp, vp, vp, f32,
                                                  i32, i32, vp]
    L.orbx_search_by_projection_mappoints_fisheye.argtypes = [vp, C.POINTER(FrameDesc), vp, i32, vp, vp, vp, i32] + [vp] * 12 + [f32, f32, vp]
    L.orbx_search_by_projection_frame_fisheye.argtypes = [vp, C.POINTER(FrameDesc), vp, i32, vp, i32] + [vp] * 8 + [f32, i32, i32, vp]
    L.orbx_search_by_bow_frame_fisheye.argtypes = [vp, vp, vp, vp, i32, C.POINTER(FeatVec), vp, vp, i32, i32, C.POINTER(FeatVec), f32, i32, vp]
    L.orbx_undistort_keypoints.argtypes = [vp, C.POINTER(Camera), vp, i32, vp]
    L.orbx_image_bounds.argtypes = [C.POINTER(Camera), i32, i32, vp]
    L.orbx_is_in_frustum.argtypes = [vp, C.POINTER(Camera), C.POINTER(FramePose), vp, f32, i32, f32, i32] + [vp] * 11
    L.orbx_is_in_frustum_checks.argtypes = [vp, vp, i32, vp, f32, i32, f32, i32] + [vp] * 10
    L.orbx_frustum_batch_device.argtypes = [vp, C.POINTER(Camera), C.POINTER(FramePose), i32, vp, f32, i32] + [vp] * 11
    L.orbx_set_camera.argtypes = [vp, C.POINTER(Camera)]
    L.orbx_batch_download_keypoints_un.argtypes = [vp, i32, vp, i32, vp]
    L.orbx_match_consecutive_device.argtypes = [vp, f32, f32, f32, i32, vp, vp]
    L.orbx_stereo_batch_device.argtypes = [vp, vp, f32, f32]
    L.orbx_vocabulary_create.argtypes = [i32, i32, i32, vp, vp, vp, vp, C.POINTER(vp)]
    L.orbx_vocabulary_destroy.argtypes = [vp]
    L.orbx_bow_transform.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    L.orbx_distinctive_descriptors.argtypes = [vp, vp, vp, i32, vp]
    L.orbx_fuse_search.argtypes = [vp, C.POINTER(FrameDesc), vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp]
    L.orbx_stereo_batch_download.argtypes = [vp, i32, vp, vp, vp, vp]
    L.orbx_stereo_batch_download_all.argtypes = [vp, vp, vp, vp]
    L.orbx_stereo_batch_download_async.argtypes = [vp, vp, vp, vp]
    L.orbx_stereo_download_wait.argtypes = [vp]
    L.orbx_search_mappoints_batch_device.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, sz, f32, f32, vp, vp]
    L.orbx_search_by_projection_window.argtypes = [vp, C.POINTER(FrameDesc), vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, f32,
                                                   i32, vp]
    L.orbx_search_for_initialization.argtypes = [vp, vp, vp, i32, C.POINTER(FrameDesc), vp, i32, f32, i32, vp]
    fvp = C.POINTER(FeatVec)
    L.orbx_search_by_bow_frame.argtypes = [vp, vp, vp, vp, i32, fvp, vp, vp, i32, fvp, f32, i32, vp]
    L.orbx_search_by_bow_keyframes.argtypes = [vp, vp, vp, vp, i32, fvp, vp, vp, vp, i32, fvp, f32, i32, vp]
    L.orbx_search_for_triangulation.argtypes = [vp, vp, vp, vp, i32, fvp, vp, vp, vp, i32, fvp, i32, PAIR_PREDICATE, vp, vp]
    L.orbx_search_for_triangulation_pinhole.argtypes = [vp, vp, vp, i32, fvp, vp, vp, i32, fvp, i32, C.POINTER(PinholeGate), vp]
    L.orbx_search_for_triangulation_kb8.argtypes = [vp, vp, vp, i32, fvp, vp, vp, i32, fvp, i32, C.POINTER(Kb8GateStruct), vp]
    L.orbx_debug_kb8_epipolar.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    _lib = L
    return L


def check(status: int, where: str) -> int:
    if status < 0:
        raise OrbxError(status, where)
    return status


def ptr(a):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)
