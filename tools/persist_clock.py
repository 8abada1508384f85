"""Development build only (SRMAP_DEV_INSTANCES): per-wave phase stamps of the persistent tile kernel's 5th iteration
(s_memtime at the phase boundaries) on the cfg2 geometry -> mean cycles per phase and wave:  python tools/persist_clock.py"""
import ctypes, os, sys
import numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))
import srmap as sr
W, s, K = 2048, 4, 16
shifts = [[k % s, (k // s) % s] for k in range(K)]
ctx = sr.Context(0)
p = sr.Problem(ctx, W, W, 1, K, s, shifts, 3, 1.0, sr.F64)
y = torch.rand((K, 1, W // s, W // s), dtype=torch.float64, device="cuda")
x = torch.rand((1, W, W), dtype=torch.float64, device="cuda"); g = torch.empty_like(x)
p.set_observations_device(y.data_ptr())
r = p.add_regularizer(sr.REG_BTV, 0.01, 3, 0.5)
p.update_irls_weights_device(r, x.data_ptr())
p.set_impl(sr.IMPL_PERSIST)
for _ in range(200): p.eval_device(x.data_ptr(), g.data_ptr(), sr.TERM_ALL)
NWG, NW = 256, 8
dbg = torch.zeros((NWG * NW, 16), dtype=torch.int64, device="cuda")
lib = sr.load()
lib.srmap_dev_set_persist_dbg.argtypes = [ctypes.c_void_p]
lib.srmap_dev_set_persist_dbg(ctypes.c_void_p(dbg.data_ptr()))
p.eval_device(x.data_ptr(), g.data_ptr(), sr.TERM_ALL)
torch.cuda.synchronize()
lib.srmap_dev_set_persist_dbg(ctypes.c_void_p(0))
d = dbg.cpu().numpy().reshape(NWG, NW, 16).astype(np.float64)
ok = d[:, :, 8] > 0
names = ["stage (wait + x -> LDS + copies)", "barrier A", "request burst", "phase 1 data", "phase 1 reg", "barrier B", "phase 2", "partials"]
print("persistent tile kernel, iteration 5 of every workgroup: mean cycles per phase (columns = wave 0..7)")
for k, n in enumerate(names):
    dt = d[:, :, k + 1] - d[:, :, k]
    row = [dt[:, w][ok[:, w]].mean() for w in range(NW)]
    print("%-34s" % n + " ".join("%6.0f" % v for v in row) + "   mean %6.0f" % np.mean(row))
tot = d[:, :, 8] - d[:, :, 0]
print("%-34s" % "top .. end of tile" + " ".join("%6.0f" % tot[:, w][ok[:, w]].mean() for w in range(NW)))
