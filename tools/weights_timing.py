"""IRLS weight pass (w = 1 / max(1e-5, BTV values of x), irls_map_solver.cpp:128-143) at cfg2 geometry: time per call and
a fingerprint of the weights.   python tools/weights_timing.py [--hr 2048] [--reps 200]
The pass runs once per IRLS round inside srmap_solve; here through srmap_update_irls_weights_device."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))
import torch
import bench, srmap


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hr", type=int, default=2048); ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--dtype", default="f64")
    a = ap.parse_args()
    S, K, W = 4, 16, a.hr
    shifts = [(k % S, (k // S) % S) for k in range(K)]
    ctx = srmap.Context(0)
    f64 = a.dtype == "f64"
    prob = srmap.Problem(ctx, W, W, 1, K, S, shifts, 3, 1.0, srmap.F64 if f64 else srmap.F32)
    prob.add_regularizer(srmap.REG_BTV, 0.01, 3, 0.5)
    gt = bench.synth_ground_truth(W, W, 1)
    rng = np.random.default_rng(5)
    xh = gt + 0.01 * rng.standard_normal(gt.shape)
    x = torch.tensor(xh, dtype=torch.float64 if f64 else torch.float32, device="cuda:0")
    st = torch.cuda.Stream()  # the library launches on the stream it is handed: the events go on the same one
    with torch.cuda.stream(st):
        for _ in range(10):
            prob.update_irls_weights_device(0, x.data_ptr(), stream=st.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(a.reps):
            prob.update_irls_weights_device(0, x.data_ptr(), stream=st.cuda_stream)
        e1.record(st); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / a.reps
    # fingerprint through one evaluation with these weights (cost + gradient sums depend on every weight)
    g = torch.zeros_like(x)
    cost = prob.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_REG, want_cost=True)
    gd = g.double()
    print({"hr": W, "dtype": a.dtype, "us_per_pass": round(us, 2), "bytes_moved_MB": round(2 * x.numel() * x.element_size() / 1e6, 1),
           "reg_cost": repr(float(cost)), "gsum": repr(float(gd.sum())), "gabs": repr(float(gd.abs().sum()))})


if __name__ == "__main__":
    main()
