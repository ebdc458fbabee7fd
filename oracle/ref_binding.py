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


# ---- vendored DBoW2 of the reference (libdbow2_ref.so) ----
_dbow = None


def dbow_available() -> bool:
    return (_DIR / "libdbow2_ref.so").exists()


def _dbow_lib():
    global _dbow
    if _dbow is None:
        C.CDLL(str(_DIR.parent / "liborb_oracle.so"), mode=C.RTLD_GLOBAL)
        L = C.CDLL(str(_DIR / "libdbow2_ref.so"))
        vp, i32 = C.c_void_p, C.c_int
        L.dbowref_load_text.restype = vp
        L.dbowref_load_text.argtypes = [C.c_char_p]
        L.dbowref_destroy.argtypes = [vp]
        L.dbowref_size.restype = i32
        L.dbowref_size.argtypes = [vp]
        L.dbowref_transform.restype = i32
        L.dbowref_transform.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp]
        _dbow = L
    return _dbow


def write_vocabulary_text(path, k, L, child_ptr, child_idx, node_desc, word_id):
    """ORBvoc.txt format of TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1338-1431): header 'k L scoring
    weighting', then one line per non-root node in id order: parent, is-leaf, 32 descriptor bytes, weight.  Requires the
    flattened tree to number nodes so that every child list is ascending and contiguous in id order, and word ids to follow
    leaf id order (what the loader produces)."""
    n = len(word_id)
    parent = np.full(n, -1, np.int64)
    for i in range(n):
        for c in child_idx[child_ptr[i]:child_ptr[i + 1]]:
            parent[c] = i
    lines = [f"{k} {L} 0 0"]   # scoring L1_NORM, weighting TF_IDF (irrelevant for word / node ids)
    for i in range(1, n):
        lines.append(f"{parent[i]} {1 if word_id[i] >= 0 else 0} " + " ".join(str(int(b)) for b in node_desc[i]) + " 1.0")
    with open(path, "w") as f:
        f.write("\n".join(lines))   # no trailing newline: the loader's while(!f.eof()) loop would read one more (empty) node


class RefVocabulary:
    def __init__(self, path):
        self.h = _dbow_lib().dbowref_load_text(str(path).encode())
        if not self.h:
            raise RuntimeError("loadFromTextFile failed")

    def __del__(self):
        if getattr(self, "h", None):
            _dbow_lib().dbowref_destroy(self.h)
            self.h = None

    def words(self):
        return _dbow_lib().dbowref_size(self.h)

    def transform(self, desc, levelsup):
        d = np.ascontiguousarray(desc, np.uint8)
        n = len(d)
        word, node = np.zeros(n, np.int32), np.zeros(n, np.int32)
        fvn, fvf = np.zeros(n, np.int32), np.zeros(n, np.int32)
        k = _dbow_lib().dbowref_transform(self.h, d.ctypes.data, n, levelsup, word.ctypes.data, node.ctypes.data, fvn.ctypes.data, fvf.ctypes.data)
        return word, node, fvn[:k], fvf[:k]
