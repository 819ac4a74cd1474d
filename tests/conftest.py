import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN_DIR, "mppi_golden.npz"))


@pytest.fixture(scope="session")
def kat():
    import json
    with open(os.path.join(GOLDEN_DIR, "mppi_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture()
def hip_runtime():
    """ctypes handle of the HIP runtime the ENGINE is bound to.  torch wheels bundle their own libamdhip64.so: a
    process that creates an engine first and imports torch later maps two runtimes that cannot see each other's
    memory (`ctypes.CDLL("libamdhip64.so")` would then pick torch's copy) -- so look the engine's one up."""
    import ctypes
    from motion_planning_amd import _capi
    _capi.load()
    paths = sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l))
    assert paths, "no HIP runtime mapped"
    mine = [p for p in paths if os.sep + "torch" + os.sep not in p] or paths
    return ctypes.CDLL(mine[0])
