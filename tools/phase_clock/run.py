"""cfg2 evaluation through the timing-only library of build.sh: mean cycles per wave and phase of k_eval_z
(s_memtime, deferred stores).  Cycle counters of different XCDs are not synchronised: only per-wave differences are used."""
import os, sys, time, ctypes
import numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))
import srmap
srmap.LIB_PATH = os.path.join(ROOT, "tools", "phase_clock", "libsrmap_time.so")
W = 2048; s, K = 4, 16
SUBPIX = "--subpixel" in sys.argv   # the sub-pixel instance k_eval_z<...,SP> (tools/subpixel_timing.py's shifts)
shifts = [[k % s, (k // s) % s] for k in range(K)]
if SUBPIX:
    rng = np.random.default_rng(5)
    shifts = [[k % s + float(np.round(rng.uniform(-.5, .5) * 32) / 32), (k // s) % s + float(np.round(rng.uniform(-.5, .5) * 32) / 32)] for k in range(K)]
ctx = srmap.Context(0)
p = srmap.Problem(ctx, W, W, 1, K, s, shifts, 3, 1.0, srmap.F64)
y = torch.rand((K, 1, W // s, W // s), dtype=torch.float64, device="cuda")
x = torch.rand((1, W, W), dtype=torch.float64, device="cuda"); g = torch.empty_like(x)
p.set_observations_device(y.data_ptr())
r = p.add_regularizer(srmap.REG_BTV, 0.01, 3, 0.5)
p.update_irls_weights_device(r, x.data_ptr())
for _ in range(20): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
torch.cuda.synchronize()
nb = 8 * 256
buf = torch.zeros((nb, 8, 16), dtype=torch.int64, device="cuda")
L = srmap.load()
L.srmap_dbg_set_timing.argtypes = [ctypes.c_void_p]
assert L.srmap_dbg_set_timing(ctypes.c_void_p(buf.data_ptr())) == 0
p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
torch.cuda.synchronize()
L.srmap_dbg_set_timing(ctypes.c_void_p(0))
t = buf.cpu().numpy().astype(np.float64)[:, :, :10]  # [block][wave][stamp]
names = ["issue x loads", "own y prefetch", "halo y prefetch", "w loads etc", "wait loads + LDS store + barrier", "phase1 data", "phase1 reg", "barrier 2", "phase2 + store"]
d = np.diff(t, axis=2)
print("per-wave mean cycles by phase and wave index (columns = wave 0..7):")
for i, nme in enumerate(names):
    print("%-36s" % nme, " ".join("%6.0f" % v for v in d[:, :, i].mean(axis=0)), "  mean %.0f" % d[:, :, i].mean())
print("wave life mean %.0f" % (t[:, :, 9] - t[:, :, 0]).mean())
life = (t[:, :, 9].max(axis=1) - t[:, :, 0].min(axis=1)).reshape(8, 256)  # [by][bx]
print("block life by tile column:", " ".join("%6.0f" % v for v in life.mean(axis=1)))
inner = d.reshape(8, 256, 8, 9)[2:6].reshape(-1, 8, 9)
print("interior tile columns only:")
for i, nme in enumerate(names):
    print("%-36s" % nme, " ".join("%6.0f" % v for v in inner[:, :, i].mean(axis=0)), "  mean %.0f" % inner[:, :, i].mean())
