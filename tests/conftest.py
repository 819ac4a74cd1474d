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
