"""Randomised parity sweep of the HIP path against the CPU oracle: random geometries (several tiles, partial tiles),
integer and sub-pixel shifts, blur on / off, 0-2 regularisers of every kind with random IRLS weights, ties in x
(sgn(0)).  f64: 1e-11 relative on gradient and cost; f32: 1e-4 / 2e-5.  Seeds are fixed: the sweep is a regression
test, not a lottery; `SRMAP_FUZZ_CASES` raises the count for a longer hunt."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "super-resolution_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,seed", [("f64", 1), ("f64", 2), ("f32", 3)])
@pytest.mark.parametrize("subpixel", [False, True])
def test_fuzz_against_oracle(dtype, seed, subpixel):
    import oracle as orc
    import srmap
    cases = int(os.environ.get("SRMAP_FUZZ_CASES", "40"))
    f32 = dtype == "f32"
    tol_g, tol_f = (1e-4, 2e-5) if f32 else (1e-11, 1e-11)
    rng = np.random.default_rng(seed + (100 if subpixel else 0))
    ctx = srmap.Context(0)
    worst, tiled = 0.0, 0
    for it in range(cases):
        s = int(rng.integers(2, 5))
        h, w = int(rng.integers(3, 40)), int(rng.integers(3, 90))
        H, W = h * s, w * s
        C = int(rng.integers(1, 4))
        K = int(rng.integers(1, 21))
        span = int(rng.integers(0, 7))
        if subpixel:
            shifts = [[float(np.round(v * 32) / 32) for v in rng.uniform(-span - 0.5, span + 0.5, 2)] for _ in range(K)]
            shifts[0] = [float(int(shifts[0][0])), float(int(shifts[0][1]))]  # mixed: one integer frame
        else:
            shifts = [[int(v) for v in rng.integers(-span, span + 1, 2)] for _ in range(K)]
        b = int(rng.choice([0, 3]))
        sigma = float(rng.uniform(0.6, 1.6)) if b else 0.0
        regs = []
        for _ in range(int(rng.integers(0, 3))):
            kind = int(rng.choice([srmap.REG_TV, srmap.REG_TV3D, srmap.REG_BTV]))
            regs.append((kind, float(rng.uniform(0.005, 0.05)), int(rng.integers(1, 4)), float(rng.uniform(0.3, 1.0))))
        model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=b, blur_sigma=sigma)
        lr = rng.random((K, C, h, w))
        ref = orc.Problem(model, lr)
        p = srmap.Problem(ctx, W, H, C, K, s, shifts, b, sigma, srmap.F32 if f32 else srmap.F64)
        p.set_observations(lr)
        for kind, lam, rg, dc in regs:
            i = p.add_regularizer(kind, lam, rg, dc)
            ref.add_regularizer(kind, lam, rg, dc)
            wts = 0.5 + 2 * rng.random((C, H, W))
            p.set_irls_weights(i, wts)
            ref.set_irls_weights(i, wts)
        x = np.round(rng.random((C, H, W)) * 32) / 32   # exact ties exercise sgn(0)
        f_ref, g_ref = ref.objective(x)
        f, g = p.eval(x)
        eg = float(np.max(np.abs(np.ravel(g) - np.ravel(g_ref)) / np.maximum(1.0, np.abs(np.ravel(g_ref)))))
        ef = abs(f - f_ref) / max(1.0, abs(f_ref))
        worst = max(worst, eg, ef)
        desc = dict(s=s, W=W, H=H, C=C, K=K, shifts=shifts, b=b, regs=regs)
        assert eg <= tol_g and ef <= tol_f, ("case %d" % it, desc, eg, ef)
        try:
            p.set_impl(srmap.IMPL_TILED)
            f2, g2 = p.eval(x)
            tiled += 1
            assert abs(f2 - f) <= tol_f * max(1.0, abs(f)) and float(np.max(np.abs(g2 - g))) <= tol_g * max(1.0, float(np.max(np.abs(g)))), desc
        except srmap.SrmapError:
            pass  # geometry outside the tile kernel: AUTO took the direct kernels
    assert tiled >= cases // 4, "the sweep barely reached the tile kernel (%d of %d)" % (tiled, cases)
