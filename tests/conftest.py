import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "train-procgen-pytorch_amd")
GOLD = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_npz(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def npz_json(z, key):
    return json.loads(bytes(z[key]).decode())


def npz_params(z, prefix="p/"):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


@pytest.fixture(scope="session")
def gold():
    return GOLD
