"""Row-band (spatial) sharding of the objective: every band
problem lives on its owned HR rows plus halo rows, counts only the cost of the
rows it owns (srmap_problem_set_cost_rows) and returns the gradient of every
row; owned gradients stitched together and owned costs summed must equal the
single-problem evaluation.  One GPU emulates the ranks one after the other
(the multi-process orchestration is tests/test_distributed_cpu.py)."""
import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu

TOL = {0: 1e-12, 1: 2e-5}


from parity_log import relerr  # noqa: E402  (max |a - ref| / max(1, |ref|), logged when SRMAP_PARITY_LOG is set)


BAND_CASES = [
    # W, H, C, s, shifts, blur, sigma, reg, world, impl
    (64, 96, 1, 4, [[k % 4, k // 4] for k in range(16)], 3, 1.0, ("btv", 3, 0.5), 3, "auto"),
    (40, 60, 2, 2, [[0, 0], [1, 1], [-1, 2], [2, -1]], 3, 1.0, ("tv",), 2, "auto"),
    (300, 128, 1, 4, [[k % 4, (k * 3) % 4] for k in range(8)], 3, 1.0, ("btv", 3, 0.5), 4, "tiled"),
    (48, 54, 1, 3, [[0, 0], [2, 1], [-2, -1], [1, -2]], 0, 0.0, ("btv", 2, 0.7), 3, "direct"),
    (32, 48, 1, 4, [[0.5, 0.25], [-1.25, 1.5], [2.0, -0.75]], 3, 1.0, ("tv",), 2, "auto"),  # sub-pixel: direct kernels
]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", range(len(BAND_CASES)))
def test_row_bands_reassemble_the_full_objective(case, dtype):
    import srmap
    import srmap_dist
    W, H, C, s, shifts, b, sigma, reg, world, impl = BAND_CASES[case]
    ctx = srmap.Context(0)
    rng = np.random.default_rng(4200 + case)
    K = len(shifts)
    model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=b, blur_sigma=sigma)
    gt = rng.random((C, H, W))
    lr = np.stack([model.apply(gt, k) for k in range(K)]) + 0.02 * rng.standard_normal((K, C, H // s, W // s))
    x = rng.random((C, H, W))
    wts = 0.5 + rng.random((C, H, W))
    impls = {"auto": srmap.IMPL_AUTO, "direct": srmap.IMPL_DIRECT, "tiled": srmap.IMPL_TILED}

    def add_reg(p):
        if reg[0] == "btv":
            return p.add_regularizer(srmap.REG_BTV, 0.03, reg[1], reg[2])
        return p.add_regularizer(srmap.REG_TV, 0.03)

    full = srmap.Problem(ctx, W, H, C, K, s, shifts, b, sigma, dtype)
    full.set_impl(impls[impl])
    full.set_observations(lr)
    full.set_irls_weights(add_reg(full), wts)
    f_ref, g_ref = full.eval(x)

    amax = int(np.ceil(np.max(np.abs(np.asarray(shifts, dtype=float)))))
    halo = srmap_dist.band_halo(s, b, amax + 1, reg[1] if reg[0] == "btv" else 1)
    g = np.zeros_like(x)
    f = 0.0
    for rank in range(world):
        (r0, r1), (e0, e1) = srmap_dist.row_band(H, s, world, rank, halo)
        band = srmap.Problem(ctx, W, e1 - e0, C, K, s, shifts, b, sigma, dtype)
        band.set_impl(impls[impl])
        band.set_observations(lr[:, :, e0 // s:e1 // s, :])
        band.set_irls_weights(add_reg(band), wts[:, e0:e1, :])
        band.set_cost_rows(r0 - e0, r1 - e0)
        fb, gb = band.eval(x[:, e0:e1, :])
        f += fb
        g[:, r0:r1, :] = np.asarray(gb).reshape(C, e1 - e0, W)[:, r0 - e0:r1 - e0, :]
    # bands are evaluated with different tile alignments: summation order differs -> 10x the element tolerance
    assert relerr(g, g_ref) <= 10 * TOL[dtype]
    assert abs(f - f_ref) <= 10 * TOL[dtype] * max(1.0, abs(f_ref))


def test_cost_rows_argument_checks():
    import srmap
    ctx = srmap.Context(0)
    p = srmap.Problem(ctx, 16, 16, 1, 1, 4, [[0, 0]], 0, 0.0, srmap.F64)
    with pytest.raises(srmap.SrmapError):
        p.set_cost_rows(2, 8)      # not a multiple of the scale
    with pytest.raises(srmap.SrmapError):
        p.set_cost_rows(8, 8)      # empty
    with pytest.raises(srmap.SrmapError):
        p.set_cost_rows(0, 20)     # beyond the image
    p.set_cost_rows(4, 16)
