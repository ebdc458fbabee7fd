#!/usr/bin/env python3
"""bench.py end to end on the CPU SIMT emulator (tiny batch): launcher-less rank, settle / warm-up / timed loop, parity self-check,
host-input leg, JSON line.  The numbers mean nothing (seconds per step); the point is that every code path of the bench runs.
usage: python tests/simt/bench_emul.py [--workload euroc|kitti|tumvi] [more bench.py flags]"""
import os
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import orb_slam3_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = ROOT / "tests" / "simt" / "build" / "liborbx_emul.so"
import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self
torch.Tensor.pin_memory = lambda self, *a, **k: self
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.set_device = lambda *a, **k: None
torch.cuda.is_available = lambda: True
torch.cuda.device_count = lambda: 1
for _name in ("tensor", "zeros", "full", "empty"):
    def _wrap(fn):
        return lambda *a, **k: fn(*a, **{kk: vv for kk, vv in k.items() if not (kk == "device" and str(vv).startswith("cuda"))})
    setattr(torch, _name, _wrap(getattr(torch, _name)))
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511")
sys.argv = ["bench.py", "--gpus", "1", "--batch", "4", "--steps", "2", "--warmup", "1", "--settle", "1", "--settle-seconds", "0", "--cpu-frames", "0", "--no-pmc",
            "--no-profile", "--verify", "2"] + sys.argv[1:]
runpy.run_path(str(ROOT / "bench.py"), run_name="__main__")
