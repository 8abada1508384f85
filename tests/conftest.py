import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "super-resolution_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def literals():
    with open(os.path.join(GOLDEN, "reference_literals.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def fb_gray():
    import numpy as np
    with open(os.path.join(GOLDEN, "fb_gray.json")) as f:
        d = json.load(f)
    return np.array(d["data"], dtype=np.float64) / 255.0
