"""orb_slam3_amd -- MI355X (gfx950) native ORB front-end for ORB-SLAM3.

The product is liborbx.so (hand-written HIP kernels behind the C ABI of include/orbx.h).  This package is the
Python host-side mirror of the reference's ORBextractor / ORBmatcher surface over that ABI, plus the synthetic
frame generator used by tests and bench.  Importing the package does not load the GPU library; constructing an
ORBextractor / ORBmatcher does, and fails loudly when liborbx.so or a HIP device is missing (no CPU fallback).
"""
from ._lib import KP_DTYPE, OrbxError, LIB_PATH  # noqa: F401
from .extractor import ORBextractor  # noqa: F401
from .matcher import ORBmatcher, ORBVocabulary, FrameView, FeatureVector  # noqa: F401

__all__ = ["ORBextractor", "ORBmatcher", "ORBVocabulary", "FrameView", "FeatureVector", "KP_DTYPE", "OrbxError", "LIB_PATH"]
