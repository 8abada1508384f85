"""Shared error measure of the parity tests, with an optional log of every measured value.

relerr(a, ref) = max |a - ref| / max(1, |ref|)  (the per-element bar).  With SRMAP_PARITY_LOG=<file>
every call appends "<pytest node id>\t<error>" -- tools/parity_errors.py turns the log of one `pytest -m gpu` run into
profiles/rNN_parity_errors.txt (the measured figure per test, next to the bar the test enforces).
"""
import os

import numpy as np


def note(e, what=""):
    """Log an error figure measured elsewhere (torch on the GPU, scalar costs); returns it."""
    path = os.environ.get("SRMAP_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write("%s%s\t%.3e\n" % (os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], (" " + what) if what else "", float(e)))
    return float(e)


def relerr(a, ref):
    a, ref = np.asarray(a, dtype=float).ravel(), np.asarray(ref, dtype=float).ravel()
    e = float(np.max(np.abs(a - ref) / np.maximum(1.0, np.abs(ref)))) if a.size else 0.0
    path = os.environ.get("SRMAP_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write("%s\t%.3e\n" % (os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], e))
    return e
