"""Host-side file formats either side of the path: the
reference's hyperspectral loader tests (test/test_hyperspectral_data_loader.cpp)
restated in tests/cpp/host_io_test.cpp against the drop-in C++ loader, on the
reference's own data files (tests/golden/envi).  CPU only."""
import os
import subprocess

import numpy as np

from conftest import ROOT

GOLDEN = os.path.join(ROOT, "tests", "golden", "envi")


def test_envi_fixture_is_the_documented_cube():
    cube = np.fromfile(os.path.join(GOLDEN, "example_envi_data"), dtype="<f4").reshape(10, 9, 5)
    b, r, c = np.meshgrid(np.arange(10), np.arange(9), np.arange(5), indexing="ij")
    assert np.allclose(cube, b + r / 10.0 + c / 100.0, atol=1e-6)


def test_cpp_envi_loader_reference_cases(tmp_path):
    import __graft_entry__ as ge
    ge.build_lib()
    exe = ge.build_host_io()
    assert exe and os.path.exists(exe)
    out = subprocess.run([exe, GOLDEN, str(tmp_path)], capture_output=True, text=True, timeout=120)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0
    assert "HOST IO TESTS PASSED" in out.stdout
    # the file the C++ writer produced is plain little-endian float32 BSQ of the selected sub-cube
    saved = np.fromfile(os.path.join(str(tmp_path), "hs_data_loader_envi_out"), dtype="<f4").reshape(5, 6, 3)
    b, r, c = np.meshgrid(np.arange(5, 10), np.arange(2, 8), np.arange(0, 3), indexing="ij")
    assert np.allclose(saved, b + r / 10.0 + c / 100.0, atol=1e-6)
