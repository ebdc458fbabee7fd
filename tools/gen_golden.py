#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (regression pins; the reference ships no golden vectors)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import oracle_binding as ob  # noqa: E402
from orb_slam3_amd import synth  # noqa: E402

out = ROOT / "tests" / "golden"
out.mkdir(parents=True, exist_ok=True)
img = synth.make_test_image(5, 320, 240)
ex = ob.OracleExtractor(500, 1.2, 8, 20, 7, flags=ob.FLAG_DESC_FMA)
mono, kps, desc = ex.extract(img, lap=(0, 1000))
ex2 = ob.OracleExtractor(500, 1.2, 8, 20, 7, flags=0)
_, kps2, desc2 = ex2.extract(img, lap=(0, 1000))
assert kps.tobytes() == kps2.tobytes()
np.savez_compressed(out / "extract_320x240_seed5.npz", image=img, mono=mono, kps=kps, desc_fma=desc, desc_strict=desc2,
                    level3=ex.level_padded(3), blur2=ex.level_blurred(2))
rng = np.random.default_rng(7)
t = rng.integers(0, 256, (300, 32), dtype=np.uint8)
q = t[rng.integers(0, 300, 120)] ^ (rng.random((120, 32)) < 0.08).astype(np.uint8)
idx, dist = ob.knn2(q, t)
np.savez_compressed(out / "match_seed7.npz", q=q, t=t, knn_idx=idx, knn_dist=dist)
print("golden written:", [p.name for p in out.glob("*.npz")], "n_kps", len(kps), "desc fma!=strict rows", int((desc != desc2).any(axis=1).sum()))
