"""Runs the reference's vendored ALGLIB 3.10.0 nonlinear CG (compiled from
/root/reference by oracle/Makefile into oracle/_ref/libalglib_ref.so) on two
small objectives and stores the trajectories as tests/golden/cg_trajectories.json:
every point reported through the xupdated callback, its cost, the final x,
iteration count, nfev and termination type.  Inputs are seeded and stored too,
so the fixture is self-contained on machines without /root/reference.

    python tests/golden/make_cg_trajectories.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402


def quad_problem():
    rng = np.random.default_rng(7)
    B = rng.standard_normal((16, 16))
    A = B @ B.T + 0.5 * np.eye(16)
    b = rng.standard_normal(16)
    return A, b


def main():
    assert orc.have_ref(), "oracle/_ref/libalglib_ref.so missing (needs /root/reference)"
    out = {}
    A, b = quad_problem()

    def quad(x):
        return 0.5 * x @ A @ x - b @ x, A @ x - b

    for name, kw in (("quadratic16_default", dict(epsg=1e-6, epsf=1e-6, epsx=1e-6, maxits=50)),
                     ("quadratic16_tight", dict(epsg=1e-10, epsf=0.0, epsx=0.0, maxits=200))):
        trace = []
        x, rep = orc.mincg(quad, np.zeros(16), use_alglib=True, trace=trace, **kw)
        out[name] = {"A": A.tolist(), "b": b.tolist(), "x0": [0.0] * 16, "opts": kw,
                     "trace_f": [t[1] for t in trace], "trace_x": [t[0].tolist() for t in trace],
                     "x": x.tolist(), "iterations": rep.iterations, "nfev": rep.nfev,
                     "termination_type": rep.termination_type, "f": rep.f}

    # 8x8 HR, scale 2, 4 frames, blur 3/1.0, TV lambda 0.05 with non-unit weights
    rng = np.random.default_rng(11)
    gt = rng.random((1, 8, 8))
    shifts = [[0, 0], [1, 1], [0, 1], [1, 0]]
    model = orc.ImageModel(scale=2, shifts=shifts, blur_ksize=3, blur_sigma=1.0)
    lr = np.stack([model.apply(gt, k) for k in range(4)])
    prob = orc.Problem(model, lr)
    prob.add_regularizer(orc.REG_TV, 0.05)
    w = 0.5 + rng.random((1, 8, 8))
    prob.set_irls_weights(0, w)
    x0 = rng.random(64)
    trace = []
    kw = dict(epsg=1e-6, epsf=1e-6, epsx=1e-6, maxits=50)
    x, rep = orc.mincg(lambda v: prob.objective(v), x0, use_alglib=True, trace=trace, **kw)
    out["tv_toy_8x8"] = {"gt": gt.tolist(), "shifts": shifts, "scale": 2, "blur": [3, 1.0],
                         "lambda": 0.05, "weights": w.tolist(), "x0": x0.tolist(), "opts": kw,
                         "trace_f": [t[1] for t in trace], "x": x.tolist(),
                         "iterations": rep.iterations, "nfev": rep.nfev,
                         "termination_type": rep.termination_type, "f": rep.f}
    with open(os.path.join(HERE, "cg_trajectories.json"), "w") as f:
        json.dump(out, f)
    for k, v in out.items():
        print(k, v["iterations"], v["nfev"], v["termination_type"], v["f"])


if __name__ == "__main__":
    main()
