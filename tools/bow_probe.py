#!/usr/bin/env python3
"""SearchByBoW(KF, F) single-call latency (bench.py's case), 300 calls; ORBX_BOW_DEBUG cuts the kernel short (diagnostic)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import orb_slam3_amd as osa
from orb_slam3_amd import synth
rng = np.random.default_rng(77)
canvas1 = synth.make_canvas(1)
ex2 = osa.ORBextractor(1000, 1.2, 8, 20, 7)
_, k0, d0 = ex2(synth.frame_from_canvas(canvas1, 0, 752, 480, 1000), None, (0, 1000))
_, k1, d1 = ex2(synth.frame_from_canvas(canvas1, 1, 752, 480, 1001), None, (0, 1000))
def nodes(k, n_nodes):
    return (np.floor(k["x"] / 60).astype(np.int64) * 7 + np.floor(k["y"] / 60).astype(np.int64) * 13 + k["octave"] * 31) % n_nodes
na, nb = nodes(k0, 100), nodes(k1, 100)
fva, fvb = osa.FeatureVector.from_node_of_feature(na), osa.FeatureVector.from_node_of_feature(nb)
valid0 = (rng.random(len(k0)) < 0.7).astype(np.uint8)
m5 = osa.ORBmatcher(0.7, True)
a0, a1 = np.ascontiguousarray(k0["angle"]), np.ascontiguousarray(k1["angle"])
for _ in range(20):
    m5.SearchByBoWFrame(d0, a0, valid0, fva, d1, a1, fvb)
ts = []
for _ in range(300):
    t0 = time.perf_counter(); r = m5.SearchByBoWFrame(d0, a0, valid0, fva, d1, a1, fvb); ts.append(time.perf_counter() - t0)
print("median us %.1f  min %.1f  matches %d" % (np.median(ts) * 1e6, np.min(ts) * 1e6, r[0]))
