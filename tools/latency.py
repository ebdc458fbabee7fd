#!/usr/bin/env python3
"""Single-frame latency of the drop-in call (host image in, keypoints + descriptors out), the way Frame::ExtractORB uses it."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import orb_slam3_amd as osa
from orb_slam3_amd import synth

frames = synth.make_frames(10, 8, 752, 480)
ex = osa.ORBextractor(1000, 1.2, 8, 20, 7)
for i in range(20):
    ex(frames[i % 8], None, (0, 0))
ts = []
for i in range(300):
    t0 = time.perf_counter()
    m, k, d = ex(frames[i % 8], None, (0, 0))
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print(f"orbx_extract 752x480 nFeatures=1000: median {np.median(ts):.3f} ms  p10 {np.percentile(ts,10):.3f}  p90 {np.percentile(ts,90):.3f}  ({len(k)} keypoints)")
