"""A/B timing of measurement builds (tools/exp_build.sh):  python tools/exp_run.py name1 name2 ...
Each gpurun_ab/<name>/libsrmap.so is loaded in its own process (SRMAP_LIB), times the cfg2 evaluation (f64, 2048^2, 16
frames, blur 3, BTV(3, .5)) in a tight loop after a clock ramp, and prints the cost and a fingerprint of the gradient so
that a variant that changes the RESULT is seen at once (`product` = the library in super-resolution_amd/lib).
   --rounds N   interleaved repetitions of the whole list (default 2): variants are compared at equal clocks"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
import numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.join(%r, "super-resolution_amd", "python"))
import srmap
W, s, K = 2048, 4, 16
shifts = [[k %% s, (k // s) %% s] for k in range(K)]
ctx = srmap.Context(0)
p = srmap.Problem(ctx, W, W, 1, K, s, shifts, 3, 1.0, srmap.F64)
gen = torch.Generator(device="cuda"); gen.manual_seed(7)
y = torch.rand((K, 1, W // s, W // s), dtype=torch.float64, device="cuda", generator=gen)
x = torch.rand((1, W, W), dtype=torch.float64, device="cuda", generator=gen); g = torch.empty_like(x)
p.set_observations_device(y.data_ptr())
r = p.add_regularizer(srmap.REG_BTV, 0.01, 3, 0.5)
p.update_irls_weights_device(r, x.data_ptr())
for _ in range(3000): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
best = 1e9
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 2000
    for _ in range(n): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
    torch.cuda.synchronize(); best = min(best, 1e6 * (time.perf_counter() - t0) / n)
c = p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL, want_cost=True)
torch.cuda.synchronize()
idx = torch.arange(g.numel(), device="cuda", dtype=torch.float64).reshape(g.shape)
print("RESULT %%.2f us  cost %%.15e  gsum %%.15e  gfp %%.15e" %% (best, c, g.sum().item(), (g * torch.cos(idx)).sum().item()))
''' % ROOT
names = [a for a in sys.argv[1:] if not a.startswith("--")]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 2
if "--rounds" in sys.argv: names.remove(sys.argv[sys.argv.index("--rounds") + 1])
for rd in range(rounds):
    for n in names:
        env = dict(os.environ)
        if n != "product": env["SRMAP_LIB"] = os.path.join(ROOT, "gpurun_ab", n, "libsrmap.so")
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        print("%-22s %s" % (n, line[0][7:] if line else "FAILED: " + out.stderr[-400:]), flush=True)
