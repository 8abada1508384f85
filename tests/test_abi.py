"""CPU-side checks of the drop-in boundary: libsrmap.so loads, exports every
symbol include/srmap.h declares, and refuses to compute without a GPU (no CPU
fallback).  No compute calls are made here."""
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "srmap.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(srmap_[a-z0-9_]+)\s*\(", text)) - {"srmap_allreduce_fn"})


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build_lib()
    import srmap
    lib = srmap.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(srmap.EXPORTED_SYMBOLS) == declared


def test_header_cites_reference_interfaces():
    text = open(os.path.join(ROOT, "include", "srmap.h")).read()
    for cite in ("image_model.cpp:86-91", "image_model.cpp:93-101", "objective_function.cpp:5-20",
                 "irls_map_solver.cpp:192-265", "map_solver.cpp:52-86", "regularizer.h"):
        assert cite in text


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import srmap
    with pytest.raises(srmap.SrmapError) as e:
        srmap.Context(0)
    assert e.value.status == srmap.EHIP


def test_product_does_not_reference_oracle():
    """The product path must not include, link or import anything under oracle/."""
    pkg = os.path.join(ROOT, "super-resolution_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".hip", ".hpp", ".h", ".cpp", ".py")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower(), (dirpath, f)
