"""Solve-level parity at BASELINE.json sizes: the whole IRLS / CG solve of the C ABI (srmap_solve, f64) against the CPU
oracle's solve (oracle/srmap_oracle.c: sro_irls_solve, driven by the reference's own ALGLIB mincg when oracle/_ref is
built, else by its restatement) -- reference: src/optimization/irls_map_solver.cpp:45-157 (the IRLS loop),
src/optimization/map_solver.cpp:16-26 (the size-scaled stopping thresholds).

  (i)  configs[0] exactly: 4 frames, 2x -> 256 x 256, TV, the shifts of the reference's motion file order
       0 0 / 1 1 / 0 1 / 1 0;
  (ii) configs[1]-class at 512 x 512 HR: 16 frames, Gaussian blur 3 / 1.0, BTV(3, 0.5), lambda 0.01, the synthetic data
       and the bilinear x0 of SURVEY.md section 8(d) (what bench.py times at 2048 x 2048).

Bar: PSNR within 0.01 dB (the north-star's tolerance), the same number of IRLS rounds, CG iterations and objective
evaluations, the final cost to 1e-9 relative (cfg2-class: met; cfg1 / TV: bounded by the oracle's own sensitivity to a
last-bit perturbation, which the test measures -- see test_cfg1_solve_matches_oracle)."""
import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sr():
    import srmap
    return srmap


@pytest.fixture(scope="module")
def ctx(sr):
    return sr.Context(0)


def _compare(sr, ctx, gt, lr, x0, s, shifts, blur, reg, impl=None, cost_tol=1e-9, x_tol=1e-7):
    K, C, h, w = lr.shape
    H, W = h * s, w * s
    model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=blur[0], blur_sigma=blur[1])
    ref = orc.Problem(model, lr)
    ref.add_regularizer(*reg)
    x_ref, rep_ref = ref.solve(x0, use_alglib=orc.have_ref())
    p = sr.Problem(ctx, W, H, C, K, s, shifts, blur[0], blur[1], sr.F64)
    if impl is not None:
        p.set_impl(impl)
    p.set_observations(lr)
    p.add_regularizer(*reg)
    x, rep = p.solve(x0)
    psnr0, psnr_ref, psnr_gpu = orc.psnr(gt, x0), orc.psnr(gt, x_ref), orc.psnr(gt, x)
    print("PSNR x0 %.4f dB | oracle %.4f dB | GPU %.4f dB; IRLS rounds %d/%d, CG iterations %d/%d, evaluations %d/%d, "
          "final cost %.12g / %.12g, max |x - x_ref| %.3e" % (
              psnr0, psnr_ref, psnr_gpu, rep_ref.irls_rounds, rep.irls_rounds, rep_ref.cg_iterations, rep.cg_iterations,
              rep_ref.nfev, rep.evaluations, rep_ref.final_cost, rep.final_cost, np.max(np.abs(x - x_ref))))
    assert abs(psnr_ref - psnr_gpu) < 0.01
    assert (rep.irls_rounds, rep.cg_iterations, rep.evaluations) == (rep_ref.irls_rounds, rep_ref.cg_iterations, rep_ref.nfev)
    assert abs(rep.final_cost - rep_ref.final_cost) <= cost_tol * abs(rep_ref.final_cost)
    assert np.max(np.abs(x - x_ref)) <= x_tol
    return psnr0, psnr_ref, psnr_gpu, (ref, x_ref, rep_ref, x, rep)


def test_cfg1_solve_matches_oracle(sr, ctx):
    """BASELINE configs[0]: 4 frames, 2x -> 256 x 256, TV (lambda 0.01), integer shifts, no blur."""
    import bench
    s, K, W, H = 2, 4, 256, 256
    shifts = [[0, 0], [1, 1], [0, 1], [1, 0]]
    gt = bench.synth_ground_truth(W, H, 1)
    model = orc.ImageModel(scale=s, shifts=shifts)
    lr = np.stack([model.apply(gt, k) for k in range(K)])
    lr = lr + (5.0 / 255.0) * np.random.default_rng(777).standard_normal(lr.shape)
    x0 = bench.bilinear_upsample(lr[0], s)
    # TV weights 1 / max(1e-5, |grad x|) span five decades on this image (flat regions), and the IRLS / CG iteration
    # amplifies a last-bit difference of one evaluation (the GPU's reduction order) by ~1e7: the ORACLE ITSELF, started
    # from x0 * (1 + 1e-15 * noise), ends 5e-9 away in cost and 3e-6 away in x (1e-14: 6e-7 and 5e-5), with the same
    # iteration counts.  So the cost / iterate bars of this case are that sensitivity, measured here, not 1e-9.
    _, _, _, (ref, x_ref, rep_ref, x, rep) = _compare(sr, ctx, gt, lr, x0, s, shifts, (0, 0.0), (orc.REG_TV, 0.01, 0, 0.0),
                                                      cost_tol=1e-6, x_tol=1e-3)
    rng = np.random.default_rng(1)
    x_p, rep_p = ref.solve(x0 * (1 + 1e-14 * rng.standard_normal(x0.shape)), use_alglib=orc.have_ref())
    own_cost, own_x = abs(rep_p.final_cost - rep_ref.final_cost), np.max(np.abs(x_p - x_ref))
    print("oracle under a 1e-14 relative perturbation of x0: cost moves %.3e, x moves %.3e; GPU vs oracle: %.3e, %.3e" % (
        own_cost, own_x, abs(rep.final_cost - rep_ref.final_cost), np.max(np.abs(x - x_ref))))
    assert abs(rep.final_cost - rep_ref.final_cost) <= 10 * own_cost and np.max(np.abs(x - x_ref)) <= 10 * own_x


@pytest.mark.parametrize("impl", ["auto", "direct"])
def test_cfg2_class_solve_matches_oracle(sr, ctx, impl):
    """BASELINE configs[1] at 512 x 512 HR.  The solve LOWERS the PSNR of the bilinear start (30.8 -> 29.7 dB here,
    35.3 -> 28.4 dB at 2048 x 2048) on the oracle exactly as on the GPU: see DESIGN.md section 5.4."""
    import bench
    s, K, W, H = 4, 16, 512, 512
    shifts = [[k % s, (k // s) % s] for k in range(K)]
    gt = bench.synth_ground_truth(W, H, 1)
    model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=3, blur_sigma=1.0)
    lr = np.stack([model.apply(gt, k) for k in range(K)])
    lr = lr + (5.0 / 255.0) * np.random.default_rng(777).standard_normal(lr.shape)
    x0 = bench.bilinear_upsample(lr[0], s)
    psnr0, psnr_ref, psnr_gpu, _ = _compare(sr, ctx, gt, lr, x0, s, shifts, (3, 1.0), (orc.REG_BTV, 0.01, 3, 0.5),
                                            impl={"auto": sr.IMPL_AUTO, "direct": sr.IMPL_DIRECT}[impl])
    assert psnr_ref < psnr0  # the reference objective's minimiser fits the noise: the drop is the reference's behaviour


@pytest.mark.parametrize("dtype", [0, 1])
def test_chained_passes_equal_host_paced_passes(sr, ctx, dtype):
    """The solver queues passes whose inputs are already on the device without waiting for the host (the first trial
    evaluation behind the direction pass, the direction pass behind the beta sums: csrc/solver.hip run_cg) and lets the
    evaluation form every trial point x = xk + stp * d itself, from the UNNORMALISED direction and the norms on the device
    (csrc/cg_norm.hpp).  With srmap_irls_options.host_paced_passes = 1 every pass waits for the host, a scaling pass
    stores the normalised direction and separate passes form the trial points: the two must agree bit for bit, in both
    arithmetic types (the element d_i is one expression wherever it is formed)."""
    import bench
    s, K, W, H = 4, 16, 256, 256
    shifts = [[k % s, (k // s) % s] for k in range(K)]
    gt = bench.synth_ground_truth(W, H, 1)
    model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=3, blur_sigma=1.0)
    lr = np.stack([model.apply(gt, k) for k in range(K)])
    lr = lr + (5.0 / 255.0) * np.random.default_rng(3).standard_normal(lr.shape)
    x0 = bench.bilinear_upsample(lr[0], s)
    out = {}
    for mode in ("1", "0"):
        p = sr.Problem(ctx, W, H, 1, K, s, shifts, 3, 1.0, dtype)
        p.set_observations(lr)
        p.add_regularizer(sr.REG_BTV, 0.01, 3, 0.5)
        opts = sr.default_irls_options()
        opts.max_num_irls_iterations = 3
        opts.max_num_solver_iterations = 30
        opts.host_paced_passes = 0 if mode == "1" else 1
        x, rep = p.solve(x0, opts)
        out[mode] = (x, rep.irls_rounds, rep.cg_iterations, rep.evaluations, rep.final_cost)
        if mode == "0":
            # the beta denominator y.dk is DERIVED from sums the solver already holds ((g.d) / (s1 s2) - g_prev.dk); the
            # host-paced beta pass also sums it directly (optimization.cpp:17700-17760): reduction-order level apart
            dev = p.selfcheck()
            assert 0.0 < dev <= (1e-12 if dtype == 0 else 1e-6), dev
    assert out["1"][1:] == out["0"][1:]
    assert np.array_equal(out["1"][0], out["0"][0])


# (scale, blur, HR width, HR height, channels, frames, regulariser, dtype): partial tiles on the right / bottom edge, every
# scale and blur size of the tile path, TV and BTV, more than one channel, offsets of both signs
FOLD_GEOMS = [
    (4, 3, 200, 136, 1, 16, (2, 0.01, 3, 0.5), 0),
    (4, 3, 328, 72, 2, 8, (2, 0.02, 2, 0.7), 0),
    (4, 1, 264, 96, 1, 16, (0, 0.01, 0, 0.0), 0),
    (3, 3, 201, 99, 1, 9, (2, 0.01, 3, 0.5), 0),
    (3, 1, 150, 150, 1, 5, (0, 0.02, 0, 0.0), 0),
    (2, 3, 130, 70, 1, 4, (2, 0.01, 1, 0.5), 0),
    (2, 1, 96, 160, 3, 4, (0, 0.01, 0, 0.0), 0),
    (4, 3, 200, 136, 1, 16, (2, 0.01, 3, 0.5), 1),
    (2, 3, 130, 70, 1, 4, (0, 0.01, 0, 0.0), 1),
]


@pytest.mark.parametrize("case", range(len(FOLD_GEOMS)))
def test_fold_equals_separate_passes_over_geometries(sr, ctx, case):
    """The evaluation that forms its own trial point (window = xk + stp * d_i, d_i from the unnormalised direction and the
    norms on the device; border blocks and edge tiles included) against the separate passes (host_paced_passes = 1), over
    the geometries of the tile path: same iterates bit for bit, same counts."""
    s, b, W, H, C, K, reg, dtype = FOLD_GEOMS[case]
    rng = np.random.default_rng(900 + case)
    shifts = [[int(rng.integers(-(s - 1), s)), int(rng.integers(-(s - 1), s))] for _ in range(K)]
    shifts[0] = [0, 0]
    w, h = W // s, H // s
    W, H = w * s, h * s
    lr = rng.random((K, C, h, w))
    x0 = rng.random((C, H, W))
    out = {}
    for paced in (0, 1):
        p = sr.Problem(ctx, W, H, C, K, s, shifts, b, 1.0 if b > 1 else 0.0, dtype)
        p.set_observations(lr)
        p.add_regularizer(*reg)
        opts = sr.default_irls_options()
        opts.max_num_irls_iterations = 2
        opts.max_num_solver_iterations = 6
        opts.host_paced_passes = paced
        x, rep = p.solve(x0, opts)
        out[paced] = (x, rep.irls_rounds, rep.cg_iterations, rep.evaluations, rep.final_cost)
    assert out[0][1:] == out[1][1:]
    assert np.array_equal(out[0][0], out[1][0])


# sub-pixel shifts (the case registration produces: motion_module.cpp:18-51 with fractional dx, dy) on the tile path:
# (scale, blur, HR width, HR height, channels, shifts, regulariser, dtype)
SP_FOLD_GEOMS = [
    (4, 3, 256, 136, 1, [[0.5, 0.25], [-1.3, 2.71], [0.01, -0.99], [3.0, -2.5], [2.0, 1.0], [-0.03125, 0.96875]], (2, 0.01, 3, 0.5), 0),
    (4, 3, 328, 72, 2, [[0.25 * k - 0.8, 1.1 - 0.35 * k] for k in range(7)], (2, 0.02, 2, 0.7), 0),
    (3, 1, 201, 99, 1, [[0.75, -0.5], [1, 1], [-2.25, 0.125], [0.3, 0.3]], (0, 0.02, 0, 0.0), 0),
    (2, 3, 130, 70, 1, [[0.5, 0.5], [-0.5, 1.5], [1.25, -1.75], [0, 0]], (2, 0.01, 1, 0.5), 0),
    (4, 3, 256, 136, 1, [[0.5, 0.25], [-1.3, 2.71], [0.01, -0.99], [3.0, -2.5]], (2, 0.01, 3, 0.5), 1),
]


def _sp_problem(sr, ctx, case, impl=None):
    s, b, W, H, C, shifts, reg, dtype = SP_FOLD_GEOMS[case]
    rng = np.random.default_rng(1700 + case)
    w, h = W // s, H // s
    W, H = w * s, h * s
    lr = rng.random((len(shifts), C, h, w))
    x0 = rng.random((C, H, W))
    p = sr.Problem(ctx, W, H, C, len(shifts), s, shifts, b, 1.0 if b > 1 else 0.0, dtype)
    if impl is not None:
        p.set_impl(impl)
    p.set_observations(lr)
    p.add_regularizer(*reg)
    return p, x0


@pytest.mark.parametrize("case", range(len(SP_FOLD_GEOMS)))
def test_subpixel_fold_equals_separate_passes(sr, ctx, case):
    """Sub-pixel plans: the FORWARD tile kernel forms the trial point x = xk + stp * d_i as it loads its window and writes
    it out, the tile kernel behind it produces g.d with the gradient (the ring pass runs ahead of it into a side buffer).
    Against host_paced_passes = 1 (a scaling pass stores d, k_axpy_out forms every trial point): same iterates bit for
    bit, same counts."""
    out = {}
    for paced in (0, 1):
        p, x0 = _sp_problem(sr, ctx, case)
        opts = sr.default_irls_options()
        opts.max_num_irls_iterations = 2
        opts.max_num_solver_iterations = 6
        opts.host_paced_passes = paced
        x, rep = p.solve(x0, opts)
        out[paced] = (x, rep.irls_rounds, rep.cg_iterations, rep.evaluations, rep.final_cost)
    assert out[0][1:] == out[1][1:]
    assert np.array_equal(out[0][0], out[1][0])


@pytest.mark.parametrize("case", [0, 2, 3])
def test_subpixel_cg_on_tiles_follows_direct_kernels(sr, ctx, case):
    """g.d out of the sub-pixel tile launch (interior from the phase tables, ring from the side buffer) against the direct
    kernels, where cost and g.d come from separate reduction passes: a CG run follows evaluation for evaluation."""
    res = {}
    for name, impl in (("tiled", sr.IMPL_TILED), ("direct", sr.IMPL_DIRECT)):
        p, x0 = _sp_problem(sr, ctx, case, impl)
        x, its, nfev, term, trace = p.cg_trace(x0, 0.0, 0.0, 0.0, 4)
        res[name] = (x, its, nfev, term, np.asarray(trace))
    (x1, i1, n1, t1, tr1), (x2, i2, n2, t2, tr2) = res["tiled"], res["direct"]
    assert (i1, n1, t1) == (i2, n2, t2) and len(tr1) == len(tr2)
    assert np.max(np.abs(tr1 - tr2) / np.maximum(1.0, np.abs(tr2))) <= 1e-11
    assert np.max(np.abs(x1 - x2)) <= 1e-9
