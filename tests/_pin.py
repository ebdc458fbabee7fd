"""Two-layer comparison against the compiled reference (oracle/_ref/*.so): live when the library is present, else against the
committed outputs of that library for the same seeded inputs (tests/golden/*.npz; regenerate with ORBX_WRITE_GOLDEN=1)."""
import os
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def flat(parts):
    out = []
    for x in parts:
        a = np.atleast_1d(np.asarray(x))
        out.append(a.view(np.int32).ravel() if a.dtype == np.float32 else a.astype(np.int32).ravel())
    return np.concatenate(out)


class Pinner:
    def __init__(self, golden_name: str, live: bool):
        self.path = GOLDEN / golden_name
        self.live = live
        self.gold = dict(np.load(self.path)) if self.path.exists() else {}
        self.fresh = {}

    def pin(self, name, oracle_out, ref_call):
        """oracle_out must equal the reference's output: computed live when the compiled reference is here, else the committed one."""
        o = flat(oracle_out)
        if self.live:
            r = flat(ref_call())
            assert np.array_equal(o, r), name
            self.fresh[name] = r
            if name in self.gold and not os.environ.get("ORBX_WRITE_GOLDEN"):
                assert np.array_equal(self.gold[name], r), f"stale golden {name}"
        else:
            assert name in self.gold, f"no golden for {name} and no compiled reference"
            assert np.array_equal(o, self.gold[name]), name

    def value(self, name, ref_call, dtype=np.float32):
        """An INPUT that only the compiled reference can produce (e.g. the F12 it derived): live, or the committed copy."""
        if self.live:
            v = np.ascontiguousarray(ref_call(), dtype)
            self.fresh[name] = flat([v])
            return v
        assert name in self.gold, f"no golden for {name} and no compiled reference"
        return self.gold[name].view(dtype) if dtype == np.float32 else self.gold[name].astype(dtype)

    def finish(self):
        if self.live and os.environ.get("ORBX_WRITE_GOLDEN"):
            np.savez_compressed(self.path, **self.fresh)
