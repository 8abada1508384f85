"""The marching evaluation kernel (csrc/kernels_zmarch.hip, SRMAP_IMPL_MARCH) through the C ABI on the GPU.

Same formulation as the tile kernel, other decomposition (one resident workgroup per CU walking a band of rows; x rows
requested one step ahead straight into LDS).  It is an opt-in implementation (SRMAP_IMPL_AUTO keeps the tiles,
profiles/r05_march.txt), so it gets its own parity tests:
  * against the CPU oracle (f64 1e-12 per element, f32 2e-5) on geometries the oracle finishes in seconds -- every
    frame-to-phase assignment, ties (sgn(0) = 0), IRLS weights, terms one by one, several strips and bands, the
    image border on all four sides;
  * against the tile kernel at the full cfg2 size (the gradient is bit-equal: same arithmetic, same order);
  * what it does not cover is refused, not silently rerouted.
"""
import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu

TOL = {0: 1e-12, 1: 2e-5}


from parity_log import relerr  # noqa: E402  (max |a - ref| / max(1, |ref|), logged when SRMAP_PARITY_LOG is set)


@pytest.fixture(scope="module")
def sr():
    import srmap
    return srmap


@pytest.fixture(scope="module")
def ctx(sr):
    return sr.Context(0)


def phase_shifts(s, perm_seed):
    """K = s*s frames, one per pixel phase (the geometry the marching kernel covers), in a shuffled frame order."""
    K = s * s
    order = np.random.default_rng(perm_seed).permutation(K)
    return [[int(k % s), int((k // s) % s)] for k in order]


# (LR width, LR height): HR = 4x; widths are multiples of 64 LR cells (one strip each), heights of 4 (HR: 16)
GEOMS = [(64, 16), (64, 40), (128, 24), (192, 36)]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", range(len(GEOMS)))
def test_march_matches_oracle(sr, ctx, case, dtype):
    w, h = GEOMS[case]
    s, K = 4, 16
    W, H = w * s, h * s
    shifts = phase_shifts(s, 40 + case)
    rng = np.random.default_rng(700 + case)
    model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=3, blur_sigma=1.0)
    gt = rng.random((1, H, W))
    lr = np.stack([model.apply(gt, k) for k in range(K)]) + (5 / 255) * rng.standard_normal((K, 1, h, w))
    x = np.clip(gt + 0.05 * rng.standard_normal(gt.shape), 0, 1)
    x = np.round(x * 256) / 256  # ties exercise sgn(0) = 0
    ref = orc.Problem(model, lr)
    p = sr.Problem(ctx, W, H, 1, K, s, shifts, 3, 1.0, dtype)
    p.set_observations(lr)
    i = p.add_regularizer(sr.REG_BTV, 0.01, 3, 0.5)
    j = ref.add_regularizer(sr.REG_BTV, 0.01, 3, 0.5)
    wts = np.minimum(1.0 / np.maximum(1e-5, orc.reg_values(sr.REG_BTV, x, 3, 0.5)), 50.0)
    p.set_irls_weights(i, wts)
    ref.set_irls_weights(j, wts)
    p.set_impl(sr.IMPL_MARCH)
    tol = TOL[dtype]
    f_ref, g_ref = ref.objective(x)
    f, g = p.eval(x)
    assert abs(f - f_ref) <= (tol if dtype == 0 else 1e-5) * max(1.0, abs(f_ref))
    assert relerr(g, g_ref) <= 4 * tol
    fd_ref, gd_ref = ref.data_term(x)
    fd, gd = p.eval(x, sr.TERM_DATA)
    assert abs(fd - fd_ref) <= (tol if dtype == 0 else 1e-5) * max(1.0, abs(fd_ref))
    assert relerr(gd, gd_ref) <= 4 * tol
    fr, gr = p.eval(x, sr.TERM_REG)
    assert abs(fr - (f_ref - fd_ref)) <= (10 * tol if dtype == 0 else 1e-4) * max(1.0, abs(f_ref))
    assert relerr(gr, g_ref - gd_ref) <= 8 * tol
    # and the tile kernel on the same problem: same arithmetic in the same order
    p.set_impl(sr.IMPL_TILED)
    ft, gt_ = p.eval(x)
    assert np.array_equal(np.asarray(g), np.asarray(gt_))
    assert abs(f - ft) <= 1e-14 * max(1.0, abs(ft))


@pytest.mark.parametrize("dtype", [0, 1])
def test_march_equals_tiles_at_cfg2(sr, ctx, dtype):
    """cfg2 at full size on device-generated data: marching kernel against the tile kernel, bit for bit."""
    import torch
    W, s, K = 2048, 4, 16
    shifts = [[k % s, (k // s) % s] for k in range(K)]
    td = torch.float64 if dtype == 0 else torch.float32
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    y = torch.rand((K, 1, W // s, W // s), dtype=td, device="cuda", generator=gen)
    x = torch.rand((1, W, W), dtype=td, device="cuda", generator=gen)
    x = torch.round(x * 64) / 64  # ties
    out = {}
    for impl in (sr.IMPL_TILED, sr.IMPL_MARCH):
        p = sr.Problem(ctx, W, W, 1, K, s, shifts, 3, 1.0, dtype)
        p.set_impl(impl)
        p.set_observations_device(y.data_ptr())
        r = p.add_regularizer(sr.REG_BTV, 0.01, 3, 0.5)
        p.update_irls_weights_device(r, x.data_ptr())
        g = torch.full_like(x, float("nan"))
        c = p.eval_device(x.data_ptr(), g.data_ptr(), sr.TERM_ALL, want_cost=True)
        torch.cuda.synchronize()
        out[impl] = (c, g)
    (ct, gt_), (cm, gm) = out[sr.IMPL_TILED], out[sr.IMPL_MARCH]
    assert torch.equal(gt_, gm)
    assert abs(ct - cm) <= 1e-13 * abs(ct)


def test_march_refuses_what_it_does_not_cover(sr, ctx):
    """Positive frame offsets, scales other than 4, no blur: SRMAP_IMPL_MARCH reports SRMAP_EUNSUPPORTED (the caller
    asked for this kernel; SRMAP_IMPL_AUTO is how one gets "whatever covers it")."""
    rng = np.random.default_rng(3)
    for (s, b, shifts) in ((4, 3, [[-(k % 4), -((k // 4) % 4)] for k in range(16)]),   # offsets of the other sign
                           (2, 3, [[k % 2, (k // 2) % 2] for k in range(4)]),
                           (4, 0, [[k % 4, (k // 4) % 4] for k in range(16)])):
        K = len(shifts)
        w, h = 64, 16
        W, H = w * s, h * s
        p = sr.Problem(ctx, W, H, 1, K, s, shifts, b, 1.0 if b else 0.0, sr.F64)
        p.set_observations(rng.random((K, 1, h, w)))
        p.add_regularizer(sr.REG_BTV, 0.01, 3, 0.5)
        p.set_impl(sr.IMPL_MARCH)
        with pytest.raises(sr.SrmapError):
            p.eval(rng.random((1, H, W)))
        p.set_impl(sr.IMPL_AUTO)
        p.eval(rng.random((1, H, W)))
