import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # ORBX_TEST_EMULATOR=1: the `-m gpu` tests that only use the C ABI with host buffers run HERE, on the CPU, against liborbx's own
    # host + device sources compiled for the SIMT emulator (tests/simt/: python tests/simt/build.py).  A way to exercise kernel LOGIC
    # without a GPU -- it says nothing about timing, races between streams or the hardware; tests that need torch.cuda still fail.
    if os.environ.get("ORBX_TEST_EMULATOR"):
        import orb_slam3_amd._lib as _lib
        # ORBX_TEST_EMULATOR=ubsan / asan: the sanitizer builds (python tests/simt/build.py --ubsan / --asan)
        kind = os.environ["ORBX_TEST_EMULATOR"]
        emul = ROOT / "tests" / "simt" / "build" / ("liborbx_emul_%s.so" % kind if kind in ("ubsan", "asan") else "liborbx_emul.so")
        if not emul.exists():
            raise pytest.UsageError(f"{emul} is missing: run python tests/simt/build.py")
        _lib.LIB_PATH = emul
        import torch   # "device" memory is host memory under the emulator
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.Tensor.pin_memory = lambda self, *a, **k: self
        torch.cuda.synchronize = lambda *a, **k: None
        for _name in ("zeros", "full", "empty", "ones"):   # device="cuda" -> host
            def _wrap(fn):
                def f(*a, **k):
                    if str(k.get("device", "")).startswith("cuda"):
                        k.pop("device")
                    return fn(*a, **k)
                return f
            setattr(torch, _name, _wrap(getattr(torch, _name)))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_binding as ob
    ob.lib()
    return ob


@pytest.fixture(scope="session")
def canvas1():
    from orb_slam3_amd import synth
    return synth.make_canvas(1)
