"""cfg2 evaluation with its inputs coming from HBM rather than from the 256 MiB Infinity Cache: NP independent cfg2
problems (own observations, weights, x, g: 134 MB each in f64) evaluated round robin, so that a problem's working set
has been displaced by the others' when its turn comes again.  NP = 1 is bench.py's situation (same buffers every step).
   python tools/hbm_fed_timing.py [--dtype f32]"""
import os, sys, time
import numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))
import srmap
f32 = "--dtype" in sys.argv and sys.argv[sys.argv.index("--dtype") + 1] == "f32"
W = 2048; s, K = 4, 16
shifts = [[k % s, (k // s) % s] for k in range(K)]
ctx = srmap.Context(0)
td = torch.float32 if f32 else torch.float64


def make():
    p = srmap.Problem(ctx, W, W, 1, K, s, shifts, 3, 1.0, srmap.F32 if f32 else srmap.F64)
    y = torch.rand((K, 1, W // s, W // s), dtype=td, device="cuda")
    x = torch.rand((1, W, W), dtype=td, device="cuda"); g = torch.empty_like(x)
    p.set_observations_device(y.data_ptr())
    r = p.add_regularizer(srmap.REG_BTV, 0.01, 3, 0.5)
    p.update_irls_weights_device(r, x.data_ptr())
    return p, x, g, y


for NP in (1, 2, 3, 4, 6):
    ps = [make() for _ in range(NP)]
    t_end = time.perf_counter() + 0.15  # clock ramp
    while time.perf_counter() < t_end:
        for p, x, g, _ in ps: p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
        torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 600
        for i in range(n):
            p, x, g, _ = ps[i % NP]
            p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
        torch.cuda.synchronize()
        best = min(best, 1e6 * (time.perf_counter() - t0) / n)
    print("%d problem(s) round robin (%4d MB of inputs + outputs in rotation): %.1f us / evaluation" % (
        NP, NP * (67 if f32 else 134), best))
    del ps
