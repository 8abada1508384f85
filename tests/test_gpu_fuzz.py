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


@pytest.mark.parametrize("dtype,seed", [("f64", 11), ("f32", 12)])
def test_fuzz_row_bands(dtype, seed):
    """Random row-band partitions: owned gradients stitched + owned costs summed == the one-problem
    evaluation, for random geometries / shifts / regularisers, integer and sub-pixel shifts, AUTO implementation."""
    import oracle as orc  # noqa: F401  (path set-up above; the reference here is the un-sharded HIP evaluation)
    import srmap
    import srmap_dist
    cases = int(os.environ.get("SRMAP_FUZZ_CASES", "25"))
    f32 = dtype == "f32"
    tol = 2e-4 if f32 else 1e-10
    rng = np.random.default_rng(seed)
    ctx = srmap.Context(0)
    for it in range(cases):
        s = int(rng.integers(2, 5))
        h, w = int(rng.integers(12, 60)), int(rng.integers(3, 70))
        H, W = h * s, w * s
        C = int(rng.integers(1, 3))
        K = int(rng.integers(1, 10))
        span = int(rng.integers(0, 4))
        if rng.random() < 0.3:
            shifts = [[float(np.round(v * 32) / 32) for v in rng.uniform(-span - 0.5, span + 0.5, 2)] for _ in range(K)]
        else:
            shifts = [[int(v) for v in rng.integers(-span, span + 1, 2)] for _ in range(K)]
        b = int(rng.choice([0, 3]))
        sigma = float(rng.uniform(0.6, 1.6)) if b else 0.0
        kind = int(rng.choice([srmap.REG_TV, srmap.REG_BTV]))
        rg, dc = int(rng.integers(1, 4)), float(rng.uniform(0.3, 1.0))
        world = int(rng.integers(2, 5))
        lr = rng.random((K, C, h, w))
        x = np.round(rng.random((C, H, W)) * 64) / 64
        wts = 0.5 + rng.random((C, H, W))
        dt = srmap.F32 if f32 else srmap.F64
        full = srmap.Problem(ctx, W, H, C, K, s, shifts, b, sigma, dt)
        full.set_observations(lr)
        full.set_irls_weights(full.add_regularizer(kind, 0.03, rg, dc), wts)
        f_ref, g_ref = full.eval(x)
        amax = int(np.ceil(np.max(np.abs(np.asarray(shifts, dtype=float))))) if K else 0
        halo = srmap_dist.band_halo(s, b, amax + 1, rg if kind == srmap.REG_BTV else 1)
        g = np.zeros_like(x)
        f = 0.0
        desc = dict(s=s, W=W, H=H, C=C, K=K, shifts=shifts, b=b, kind=kind, rg=rg, world=world, halo=halo)
        for rank in range(world):
            (r0, r1), (e0, e1) = srmap_dist.row_band(H, s, world, rank, halo)
            if r1 <= r0:
                continue
            band = srmap.Problem(ctx, W, e1 - e0, C, K, s, shifts, b, sigma, dt)
            band.set_observations(lr[:, :, e0 // s:e1 // s, :])
            band.set_irls_weights(band.add_regularizer(kind, 0.03, rg, dc), wts[:, e0:e1, :])
            band.set_cost_rows(r0 - e0, r1 - e0)
            fb, gb = band.eval(x[:, e0:e1, :])
            f += fb
            g[:, r0:r1, :] = np.asarray(gb).reshape(C, e1 - e0, W)[:, r0 - e0:r1 - e0, :]
        eg = float(np.max(np.abs(g - g_ref) / np.maximum(1.0, np.abs(g_ref))))
        assert eg <= tol, ("case %d" % it, desc, eg)
        assert abs(f - f_ref) <= tol * max(1.0, abs(f_ref)), ("case %d" % it, desc, f, f_ref)


@pytest.mark.parametrize("seed", [21])
def test_fuzz_terms_and_implementations(seed):
    """Term subsets, cost-only calls and the direct implementation against the oracle on random problems."""
    import oracle as orc
    import srmap
    cases = int(os.environ.get("SRMAP_FUZZ_CASES", "30"))
    rng = np.random.default_rng(seed)
    ctx = srmap.Context(0)
    for it in range(cases):
        s = int(rng.integers(1, 5))
        h, w = int(rng.integers(3, 30)), int(rng.integers(3, 80))
        H, W = h * s, w * s
        C = int(rng.integers(1, 4))
        K = int(rng.integers(1, 8))
        if rng.random() < 0.4:
            shifts = [[float(np.round(v * 32) / 32) for v in rng.uniform(-3.5, 3.5, 2)] for _ in range(K)]
        else:
            shifts = [[int(v) for v in rng.integers(-4, 5, 2)] for _ in range(K)]
        b = int(rng.choice([0, 3, 5]))
        sigma = float(rng.uniform(0.6, 1.6)) if b else 0.0
        regs = [(int(rng.choice([srmap.REG_TV, srmap.REG_TV3D, srmap.REG_BTV])), float(rng.uniform(0.005, 0.05)),
                 int(rng.integers(1, 5)), float(rng.uniform(0.3, 1.0))) for _ in range(int(rng.integers(1, 3)))]
        model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=b, blur_sigma=sigma)
        lr = rng.random((K, C, h, w))
        ref = orc.Problem(model, lr)
        p = srmap.Problem(ctx, W, H, C, K, s, shifts, b, sigma, srmap.F64)
        p.set_observations(lr)
        for kind, lam, rg, dc in regs:
            i = p.add_regularizer(kind, lam, rg, dc)
            ref.add_regularizer(kind, lam, rg, dc)
            wts = 0.5 + 2 * rng.random((C, H, W))
            p.set_irls_weights(i, wts)
            ref.set_irls_weights(i, wts)
        x = np.round(rng.random((C, H, W)) * 16) / 16
        desc = dict(s=s, W=W, H=H, C=C, K=K, shifts=shifts, b=b, regs=regs)
        fd_ref, gd_ref = ref.data_term(x)
        f_ref, g_ref = ref.objective(x)
        for impl in (srmap.IMPL_AUTO, srmap.IMPL_DIRECT):
            p.set_impl(impl)
            fd, gd = p.eval(x, srmap.TERM_DATA)
            f, g = p.eval(x)
            fr, gr = p.eval(x, srmap.TERM_REG)
            fc, _ = p.eval(x, srmap.TERM_ALL, want_grad=False)
            for a, r in ((gd, gd_ref), (g, g_ref), (np.asarray(gd) + np.asarray(gr), g_ref)):
                e = float(np.max(np.abs(np.ravel(a) - np.ravel(r)) / np.maximum(1.0, np.abs(np.ravel(r)))))
                assert e <= 1e-11, ("case %d impl %d" % (it, impl), desc, e)
            assert abs(fd - fd_ref) <= 1e-11 * max(1.0, abs(fd_ref)), desc
            assert abs(f - f_ref) <= 1e-11 * max(1.0, abs(f_ref)), desc
            assert abs(fd + fr - f) <= 1e-11 * max(1.0, abs(f)) and abs(fc - f) <= 1e-12 * max(1.0, abs(f)), desc
