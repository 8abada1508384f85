"""Marching kernel (SRMAP_IMPL_MARCH) against the 8-row tile kernel (SRMAP_IMPL_TILED) on the same inputs: cost, gradient
differences (whole image / interior), and time per evaluation of both.
   python tools/march_check.py [--hr 2048] [--dtype f64|f32] [--frames 16] [--shiftmode bench|zero|neg|mixed] [--notime]"""
import os, sys, time
import numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))
import srmap
def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default
W = int(arg("--hr", 2048)); H = int(arg("--hh", W))
f32 = arg("--dtype", "f64") == "f32"
s = int(arg("--scale", 4)); K = int(arg("--frames", s * s))
mode = arg("--shiftmode", "bench")
if mode == "bench": shifts = [[k % s, (k // s) % s] for k in range(K)]
elif mode == "zero": shifts = [[0, 0] for k in range(K)]
elif mode == "neg": shifts = [[-(k % s), -((k // s) % s)] for k in range(K)]
else: shifts = [[(k * 5) % 7 - 3, (k * 3) % 5 - 2] for k in range(K)]
ctx = srmap.Context(0)
td = torch.float32 if f32 else torch.float64
gen = torch.Generator(device="cuda"); gen.manual_seed(11)
y = torch.rand((K, 1, H // s, W // s), dtype=td, device="cuda", generator=gen)
x = torch.rand((1, H, W), dtype=td, device="cuda", generator=gen)
res = {}
for name, impl in (("tiled", srmap.IMPL_TILED), ("march", srmap.IMPL_MARCH)):
    p = srmap.Problem(ctx, W, H, 1, K, s, shifts, 3, 1.0, srmap.F32 if f32 else srmap.F64)
    p.set_impl(impl)
    p.set_observations_device(y.data_ptr())
    r = p.add_regularizer(srmap.REG_BTV, 0.01, 3, 0.5)
    p.update_irls_weights_device(r, x.data_ptr())
    g = torch.full_like(x, float("nan"))
    c = p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL, want_cost=True)
    torch.cuda.synchronize()
    gd = torch.full_like(x, float("nan")); cd = p.eval_device(x.data_ptr(), gd.data_ptr(), srmap.TERM_DATA, want_cost=True)
    gr = torch.full_like(x, float("nan")); cr = p.eval_device(x.data_ptr(), gr.data_ptr(), srmap.TERM_REG, want_cost=True)
    torch.cuda.synchronize()
    t = None
    if "--notime" not in sys.argv:
        for _ in range(1500): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); n = 1500
            for _ in range(n): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
            torch.cuda.synchronize(); best = min(best, 1e6 * (time.perf_counter() - t0) / n)
        t = best
    res[name] = (c, g.clone(), cd, gd, cr, gr, t)
    print("%-6s cost %.15e  data %.15e  reg %.15e  %s" % (name, c, cd, cr, "%.2f us" % t if t else ""), flush=True)
a, b = res["tiled"], res["march"]
def cmp(tag, ga, gb):
    d = (ga - gb).abs()
    bad = torch.isnan(d)
    d = torch.where(bad, torch.full_like(d, float("inf")), d)
    sc = ga.abs().max().item()
    m = d.max().item()
    idx = int(d.argmax().item()); rr, cc = divmod(idx % (H * W), W)
    E = 40
    di = d[:, E:H - E, E:W - E].max().item()
    print("%-5s max|dg| %.3e (rel %.3e) at (%d, %d); interior %.3e; nan %d" % (tag, m, m / sc, rr, cc, di, int(bad.sum().item())))
    return m / sc
e1 = cmp("all", a[1], b[1]); e2 = cmp("data", a[3], b[3]); e3 = cmp("reg", a[5], b[5])
for tag, i in (("all", 0), ("data", 2), ("reg", 4)):
    print("cost %-4s rel diff %.3e" % (tag, abs(a[i] - b[i]) / max(1.0, abs(a[i]))))
tol = 2e-5 if f32 else 1e-12
ok = max(e1, e2, e3) < tol and all(abs(a[i] - b[i]) / max(1.0, abs(a[i])) < tol for i in (0, 2, 4))
print("PARITY", "OK" if ok else "FAIL")
