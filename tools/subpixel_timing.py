"""cfg2 geometry with SUB-PIXEL shifts (1/32-px quantised bilinear warps, SURVEY.md 8d "unpinned mode"): evaluation time.
   python tools/subpixel_timing.py [--hr 2048]"""
import os, sys, time
import numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))
import srmap

W = int(sys.argv[sys.argv.index("--hr") + 1]) if "--hr" in sys.argv else 2048
s, K = 4, 16
rng = np.random.default_rng(1)
torch.manual_seed(1)
for name, frac in (("integer", False), ("sub-pixel", True)):
    shifts = [[k % s + (np.round(rng.uniform(-.5, .5) * 32) / 32 if frac else 0),
               (k // s) % s + (np.round(rng.uniform(-.5, .5) * 32) / 32 if frac else 0)] for k in range(K)]
    ctx = srmap.Context(0)
    p = srmap.Problem(ctx, W, W, 1, K, s, shifts, 3, 1.0, srmap.F64)
    y = torch.rand((K, 1, W // s, W // s), dtype=torch.float64, device="cuda")
    x = torch.rand((1, W, W), dtype=torch.float64, device="cuda"); g = torch.empty_like(x)
    p.set_observations_device(y.data_ptr())
    r = p.add_regularizer(srmap.REG_BTV, 0.01, 3, 0.5)
    p.update_irls_weights_device(r, x.data_ptr())
    for _ in range(3): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
    torch.cuda.synchronize()
    tr = time.perf_counter()  # sustained clocks first (as bench.py)
    while time.perf_counter() - tr < 0.1:
        for _ in range(10): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
        torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 300
    for _ in range(n): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cost = p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL, want_cost=True)
    print("%-10s shifts: %.1f us / evaluation   (cost %r, sum g %r, sum |g| %r)" % (
        name, 1e6 * dt / n, float(cost), float(g.sum()), float(g.abs().sum())))
