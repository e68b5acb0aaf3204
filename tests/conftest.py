import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    p = g.load_package()
    p.lib()
    return p


@pytest.fixture(scope="session")
def orc():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def weather():
    import numpy as np
    return np.ascontiguousarray(np.loadtxt(os.path.join(ROOT, "tests", "golden", "weather_stations.csv"), delimiter=","))
