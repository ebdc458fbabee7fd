#!/usr/bin/env python3
"""Phase timing of k_octree_par for frame 0 of a 128-frame EuRoC batch (debug aid; prints microseconds per phase)."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import orb_slam3_amd as osa
from orb_slam3_amd import synth, _lib

NF = int(sys.argv[1]) if len(sys.argv) > 1 else 16
frames = synth.make_frames(10, NF, 752, 480)
d = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
ex = osa.ORBextractor(1000, 1.2, 8, 20, 7)
L = _lib.lib()
L.orbx_debug_octree_timing.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
names = ["total", "roots", "A keys", "B nodes", "C scatter", "sort", "final", "#breadth", "#sorted", "C", "max sort n", "nodes", "sort:lds-part", "sort:reg-part", "sort:rank", "#partitions"]
for level in (0, 1, 3):
    out = np.zeros(16, np.int64)
    L.orbx_debug_octree_timing(ex._h, level, None)
    for _ in range(3):
        ex.extract_batch_device(d.data_ptr(), NF, 752, 480, 752, 752 * 480, (0, 0))
        ex.sync()
    L.orbx_debug_octree_timing(ex._h, level, out.ctypes.data_as(C.c_void_p))
    print("level", level, {n: (round(out[k] / 3 / 100.0, 2) if (k < 7 or 12 <= k <= 14) else int(out[k]) if 9 <= k <= 11 else out[k] / 3) for k, n in enumerate(names)})
