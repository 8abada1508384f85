import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "super-resolution_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# On a GPU box initialise torch's HIP runtime BEFORE libsrmap.so is loaded: torch ships its own libamdhip64, and a
# process that has already initialised the system runtime through libsrmap.so makes torch report "no ROCm-capable
# device" (the full-size tests generate their data with torch on the GPU).  No effect without a GPU.
try:
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
        torch.zeros(1, device="cuda")
except Exception:  # pragma: no cover
    pass


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
