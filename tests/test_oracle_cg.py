"""The oracle's nonlinear-CG restatement (sro_mincg) against trajectories of the
reference's vendored ALGLIB 3.10.0 (tests/golden/cg_trajectories.json, made by
tests/golden/make_cg_trajectories.py) and, when oracle/_ref is present, against
ALGLIB live."""
import json
import os

import numpy as np
import pytest

import oracle as orc
from conftest import GOLDEN


@pytest.fixture(scope="module")
def traj():
    with open(os.path.join(GOLDEN, "cg_trajectories.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", ["quadratic16_default", "quadratic16_tight"])
def test_quadratic_trajectory(traj, name):
    T = traj[name]
    A, b = np.array(T["A"]), np.array(T["b"])
    trace = []
    x, rep = orc.mincg(lambda v: (0.5 * v @ A @ v - b @ v, A @ v - b), np.array(T["x0"]), trace=trace, **T["opts"])
    assert rep.iterations == T["iterations"]
    assert rep.nfev == T["nfev"]
    assert rep.termination_type == T["termination_type"]
    assert len(trace) == len(T["trace_f"])
    # same operations in the same order: bit-exact on this platform, but allow
    # libm/BLAS differences in the numpy objective across hosts
    assert np.allclose([t[1] for t in trace], T["trace_f"], rtol=1e-12, atol=1e-14)
    assert np.allclose(x, T["x"], rtol=1e-9, atol=1e-12)
    assert rep.f == pytest.approx(T["f"], rel=1e-12)


def _toy(T):
    model = orc.ImageModel(scale=T["scale"], shifts=T["shifts"], blur_ksize=T["blur"][0], blur_sigma=T["blur"][1])
    gt = np.array(T["gt"])
    lr = np.stack([model.apply(gt, k) for k in range(len(T["shifts"]))])
    prob = orc.Problem(model, lr)
    prob.add_regularizer(orc.REG_TV, T["lambda"])
    prob.set_irls_weights(0, np.array(T["weights"]))
    return prob


def test_tv_toy_trajectory(traj):
    T = traj["tv_toy_8x8"]
    prob = _toy(T)
    trace = []
    x, rep = orc.mincg(lambda v: prob.objective(v), np.array(T["x0"]), trace=trace, **T["opts"])
    assert (rep.iterations, rep.nfev, rep.termination_type) == (T["iterations"], T["nfev"], T["termination_type"])
    assert np.allclose([t[1] for t in trace], T["trace_f"], rtol=1e-12, atol=0)
    assert np.allclose(x, T["x"], rtol=1e-9, atol=1e-12)


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref/libalglib_ref.so not built")
def test_live_alglib_bit_exact(traj):
    """Restatement vs the real ALGLIB on the same objective, same process:
    identical iterates (bit for bit)."""
    rng = np.random.default_rng(3)
    B = rng.standard_normal((24, 24))
    A = B @ B.T + np.eye(24)
    b = rng.standard_normal(24)

    def rosen_like(v):  # non-quadratic, exercises the bracketing cases
        f = 0.5 * v @ A @ v - b @ v + 0.1 * np.sum(v ** 4)
        return f, A @ v - b + 0.4 * v ** 3

    for fun, x0 in ((rosen_like, rng.standard_normal(24) * 3),
                    (lambda v: _toy(traj["tv_toy_8x8"]).objective(v), rng.random(64))):
        ta, tb = [], []
        xa, ra = orc.mincg(fun, x0, trace=ta, maxits=40)
        xb, rb = orc.mincg(fun, x0, trace=tb, maxits=40, use_alglib=True)
        assert (ra.iterations, ra.nfev, ra.termination_type) == (rb.iterations, rb.nfev, rb.termination_type)
        assert np.array_equal(xa, xb)
        assert [t[1] for t in ta] == [t[1] for t in tb]
        assert ra.f == rb.f


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref/libalglib_ref.so not built")
def test_irls_solve_with_alglib_matches(literals):
    """IRLS loop driven by the real ALGLIB == driven by the restatement."""
    rng = np.random.default_rng(5)
    gt = rng.random((2, 12, 12))
    shifts = [[0, 0], [1, 0], [0, 1], [1, 1], [2, 1]]
    model = orc.ImageModel(scale=3, shifts=shifts, blur_ksize=3, blur_sigma=1.0)
    lr = np.stack([model.apply(gt, k) for k in range(5)])
    outs = []
    for use in (False, True):
        prob = orc.Problem(model, lr)
        prob.add_regularizer(orc.REG_BTV, 0.01, btv_range=3, btv_decay=0.5)
        opts = orc.default_irls_options()
        opts.max_num_irls_iterations = 3
        x, rep = prob.solve(np.full((2, 12, 12), 0.5), opts, use_alglib=use)
        outs.append((x, rep.cg_iterations, rep.nfev, rep.final_cost))
    assert outs[0][1:] == outs[1][1:]
    assert np.array_equal(outs[0][0], outs[1][0])
