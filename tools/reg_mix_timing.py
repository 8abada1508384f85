"""Evaluation time with regulariser mixes at 2048^2 x C channels (16 frames, 4x, blur 3/1): which part of a
hyperspectral evaluation (cfg5: BTV + 3-D TV) runs on the fused path and which on the direct kernels.
   python tools/reg_mix_timing.py [C]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))
import srmap


def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    s, K, W = 4, 16, 2048
    dev = torch.device("cuda", 0)
    shifts = [[k % s, (k // s) % s] for k in range(K)]
    ctx = srmap.Context(0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    y = torch.rand((K, C, W // s, W // s), generator=g, device=dev, dtype=torch.float64)
    x = torch.rand((C, W, W), generator=g, device=dev, dtype=torch.float64)
    gr = torch.empty_like(x)
    for name, regs in (("data only", []), ("BTV(3)", [(srmap.REG_BTV, 3, 0.5)]), ("TV", [(srmap.REG_TV, 0, 0.0)]),
                       ("TV3D", [(srmap.REG_TV3D, 0, 0.0)]), ("BTV(3) + TV3D", [(srmap.REG_BTV, 3, 0.5), (srmap.REG_TV3D, 0, 0.0)])):
        p = srmap.Problem(ctx, W, W, C, K, s, shifts, 3, 1.0, srmap.F64)
        p.set_observations_device(y.data_ptr())
        for kind, r, d in regs:
            p.add_regularizer(kind, 0.01, r, d)
        for _ in range(5):
            p.eval_device(x.data_ptr(), gr.data_ptr(), srmap.TERM_ALL)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            p.eval_device(x.data_ptr(), gr.data_ptr(), srmap.TERM_ALL)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("%-16s %8.1f us per evaluation, %6.1f us per channel" % (name, dt * 1e6, dt * 1e6 / C))
        del p


if __name__ == "__main__":
    main()
