"""End-to-end IRLS/CG solve at cfg2 (16 frames, 4x -> 2048^2, blur 3/1, BTV(3, .5)): wall time, evaluations, PSNR.
   python tools/solve_profile.py [--dtype f64] [--hr 2048] [--irls 2] [--cg 20]
Run under `rocprofv3 --kernel-trace --stats` to see how the time splits between the fused evaluation and the CG
vector kernels."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))
import bench, srmap


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f64"); ap.add_argument("--hr", type=int, default=2048)
    ap.add_argument("--irls", type=int, default=2); ap.add_argument("--cg", type=int, default=20)
    ap.add_argument("--host-paced", action="store_true", help="srmap_irls_options.host_paced_passes = 1 (every CG pass waits for the host)")
    ap.add_argument("--subpixel", action="store_true", help="random 1/32-px sub-pixel shifts (the registered-data case) instead of integer ones")
    a = ap.parse_args()
    S, K, W = 4, 16, a.hr
    gt = bench.synth_ground_truth(W, W, 1)
    shifts = [(k % S, (k // S) % S) for k in range(K)]
    if a.subpixel:
        r5 = np.random.default_rng(5)
        shifts = [(k % S + float(np.round(r5.uniform(-.5, .5) * 32) / 32), (k // S) % S + float(np.round(r5.uniform(-.5, .5) * 32) / 32)) for k in range(K)]
    ctx = srmap.Context(0)
    prob = srmap.Problem(ctx, W, W, 1, K, S, shifts, 3, 1.0, srmap.F64 if a.dtype == "f64" else srmap.F32)
    rng = np.random.default_rng(777)
    lr = np.stack([prob.apply(gt, k) for k in range(K)])
    lr = lr + (5.0 / 255.0) * rng.standard_normal(lr.shape)
    prob.set_observations(lr)
    prob.add_regularizer(srmap.REG_BTV, 0.01, 3, 0.5)
    x0 = bench.bilinear_upsample(lr[0], S)
    opts = srmap.default_irls_options()
    opts.max_num_irls_iterations = a.irls
    opts.max_num_solver_iterations = a.cg
    opts.host_paced_passes = 1 if a.host_paced else 0
    for attempt in range(2):  # the second solve is the steady state (no first-launch code loading)
        t0 = time.perf_counter()
        x, rep = prob.solve(x0, opts)
        dt = time.perf_counter() - t0
    mse = float(np.mean((x - gt) ** 2)); mse0 = float(np.mean((x0 - gt) ** 2))
    print({"dtype": a.dtype, "hr": W, "wall_s": round(dt, 4), "irls_rounds": rep.irls_rounds,
           "cg_iterations": rep.cg_iterations, "evaluations": rep.evaluations,
           "ms_per_evaluation_incl_cg": round(1e3 * dt / max(1, rep.evaluations), 4),
           "loop_ms": round(1e3 * rep.loop_seconds, 3), "wait_ms": round(1e3 * rep.wait_seconds, 3), "waits": rep.waits,
           "ms_per_evaluation_in_loop": round(1e3 * rep.loop_seconds / max(1, rep.evaluations), 4),
           "psnr_x0": round(-10 * np.log10(mse0), 3), "psnr": round(-10 * np.log10(mse), 3)})


if __name__ == "__main__":
    main()
