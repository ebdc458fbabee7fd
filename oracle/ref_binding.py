"""ctypes binding of oracle/_ref/liborb_ref.so: the REFERENCE's ORBextractor.cc compiled where it lies against oracle/ocv_shim
(TEST INFRASTRUCTURE ONLY; see ocv_shim/opencv2/opencv.hpp for what this pins).  available() is False when the library has
not been built (it needs /root/reference at build time; the built .so travels with gpurun)."""
import ctypes as C
from pathlib import Path

import numpy as np

from .oracle_binding import KP_DTYPE

_DIR = Path(__file__).resolve().parent / "_ref"
_PATH = _DIR / "liborb_ref.so"
_libs = {}


def available() -> bool:
    return _PATH.exists()


def lib(strict: bool = False):
    """strict=False: built like the reference (-O3, FMA contraction); strict=True: -ffp-contract=off."""
    if strict not in _libs:
        C.CDLL(str(_DIR.parent / "liborb_oracle.so"), mode=C.RTLD_GLOBAL)
        L = C.CDLL(str(_DIR / ("liborb_ref_strict.so" if strict else "liborb_ref.so")))
        vp, i32 = C.c_void_p, C.c_int
        L.orbref_create.restype = vp
        L.orbref_create.argtypes = [i32, C.c_float, i32, i32, i32]
        L.orbref_destroy.argtypes = [vp]
        L.orbref_extract.restype = i32
        L.orbref_extract.argtypes = [vp, vp, i32, i32, C.c_size_t, i32, i32, vp, vp, i32, vp]
        L.orbref_tables.argtypes = [vp, vp, vp, vp, vp, i32]
        L.orbref_level.restype = i32
        L.orbref_level.argtypes = [vp, i32, vp, vp, vp, C.c_size_t]
        _libs[strict] = L
    return _libs[strict]


class RefExtractor:
    """ORB_SLAM3::ORBextractor of the reference source tree."""

    def __init__(self, nfeatures, scale_factor, nlevels, ini_th, min_th, strict=False):
        self.L = lib(strict)
        self.nlevels = nlevels
        self.cap = 4 * nfeatures + 4096
        self.h = self.L.orbref_create(nfeatures, scale_factor, nlevels, ini_th, min_th)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orbref_destroy(self.h)
            self.h = None

    def extract(self, img, lap=(0, 0)):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int(0)
        mono = self.L.orbref_extract(self.h, img.ctypes.data, w, h, img.strides[0], int(lap[0]), int(lap[1]), kps.ctypes.data,
                                    desc.ctypes.data, self.cap, C.byref(n))
        return mono, kps[: n.value].copy(), desc[: n.value].copy()

    def tables(self):
        a = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        self.L.orbref_tables(self.h, *[x.ctypes.data for x in a], self.nlevels)
        return dict(scale=a[0], inv_scale=a[1], sigma2=a[2], inv_sigma2=a[3])

    def level_padded(self, level):
        w, h = C.c_int(0), C.c_int(0)
        self.L.orbref_level(self.h, level, C.byref(w), C.byref(h), None, 0)
        out = np.zeros((h.value + 38, w.value + 38), np.uint8)
        self.L.orbref_level(self.h, level, C.byref(w), C.byref(h), out.ctypes.data, out.size)
        return out
