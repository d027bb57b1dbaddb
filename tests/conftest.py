import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "emu")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running case")


def small_cases():
    return sorted(glob.glob(os.path.join(GOLD, "small_*.npz")))


def load_case(path):
    z = np.load(path)
    return dict(name=os.path.basename(path)[6:-4], q=int(path[-5]), img=z["img"], stream=z["stream"].tobytes(),
                rcon=z["rcon"], records=z["records"], ctu_state=z["ctu_state"])


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def hevce():
    import hevce_pkg
    return hevce_pkg.load()


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.lib()
    return _orc
