"""cfg2 geometry, integer shifts: wall time per srmap_eval_device call in a tight loop (no profiler).
   python tools/eval_timing.py [--hr 2048] [--dtype f64|f32] [--scale 4] [--frames 16]"""
import os, sys, time
import numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))
import srmap
W = int(sys.argv[sys.argv.index("--hr") + 1]) if "--hr" in sys.argv else 2048
f32 = "--dtype" in sys.argv and sys.argv[sys.argv.index("--dtype") + 1] == "f32"
s = int(sys.argv[sys.argv.index("--scale") + 1]) if "--scale" in sys.argv else 4
K = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else s * s
shifts = [[k % s, (k // s) % s] for k in range(K)]
ctx = srmap.Context(0)
td = torch.float32 if f32 else torch.float64
p = srmap.Problem(ctx, W, W, 1, K, s, shifts, 3, 1.0, srmap.F32 if f32 else srmap.F64)
y = torch.rand((K, 1, W // s, W // s), dtype=td, device="cuda")
x = torch.rand((1, W, W), dtype=td, device="cuda"); g = torch.empty_like(x)
p.set_observations_device(y.data_ptr())
r = p.add_regularizer(srmap.REG_BTV, 0.01, 3, 0.5)
p.update_irls_weights_device(r, x.data_ptr())
for _ in range(2000): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 2000
    for _ in range(n): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
    torch.cuda.synchronize()
    print("%d^2 %s scale %d, %d frames: %.1f us / evaluation" % (W, "f32" if f32 else "f64", s, K, 1e6 * (time.perf_counter() - t0) / n))
