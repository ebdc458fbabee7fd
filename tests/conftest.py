import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_binding as ob
    ob.lib()
    return ob


@pytest.fixture(scope="session")
def canvas1():
    from orb_slam3_amd import synth
    return synth.make_canvas(1)
