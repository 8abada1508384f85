"""Development build only (SRMAP_DEV_INSTANCES=1): per-wave time stamps and per-phase cycle counts of the marching kernel
on cfg2.   python tools/march_timeline.py"""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))
import srmap as sr
torch.cuda.init(); torch.zeros(1, device="cuda")
W = 2048; s, K = 4, 16
shifts = [[k % s, (k // s) % s] for k in range(K)]
ctx = sr.Context(0)
p = sr.Problem(ctx, W, W, 1, K, s, shifts, 3, 1.0, sr.F64)
y = torch.rand((K, 1, W // s, W // s), dtype=torch.float64, device="cuda")
x = torch.rand((1, W, W), dtype=torch.float64, device="cuda"); g = torch.empty_like(x)
p.set_observations_device(y.data_ptr())
r = p.add_regularizer(sr.REG_BTV, 0.01, 3, 0.5)
p.update_irls_weights_device(r, x.data_ptr())
p.set_impl(sr.IMPL_MARCH)
for _ in range(3000): p.eval_device(x.data_ptr(), g.data_ptr(), sr.TERM_ALL)
torch.cuda.synchronize()
dbg = torch.zeros((4096, 8), dtype=torch.int64, device="cuda")
lib = sr.load()
lib.srmap_dev_set_march_dbg.argtypes = [C.c_void_p]
lib.srmap_dev_set_march_dbg(dbg.data_ptr())
for _ in range(5): p.eval_device(x.data_ptr(), g.data_ptr(), sr.TERM_ALL)
torch.cuda.synchronize()
lib.srmap_dev_set_march_dbg(None)
d = dbg.cpu().numpy().astype(np.uint64)
d = d[d[:, 0] > 0]
n = len(d)
t0 = d[:, 0].min()
tick = 1e-2  # s_memrealtime: 100 MHz -> 10 ns
u32 = np.uint64(0xffffffff)
st = (d[:, 0] - t0) * tick
band = (d[:, 3] - d[:, 2]) * tick; tail = (d[:, 4] - d[:, 3]) * tick; end = (d[:, 4] - t0) * tick
dm = ((d[:, 5] >> np.uint64(4)) & np.uint64(3)).astype(int)
strip = ((d[:, 5] >> np.uint64(40)) & np.uint64(0xffff)).astype(int)
t_bcd = d[:, 1].astype(np.float64)
t_z = (d[:, 6] >> np.uint64(32)).astype(np.float64); t_p1 = (d[:, 6] & u32).astype(np.float64)
t_p2 = (d[:, 7] >> np.uint64(32)).astype(np.float64)
hw = (d[:, 7] & np.uint64(0xfffff)).astype(np.int64); xcc = ((d[:, 7] >> np.uint64(24)) & np.uint64(0xff)).astype(int)
print("waves %d, kernel span %.2f us (first start -> last end)" % (n, end.max()))
for k in range(3):
    m = dm == k
    if m.sum() == 0: continue
    tot = t_z[m] + t_p1[m] + t_p2[m] + t_bcd[m]
    print("dm=%d n=%4d start %.2f..%.2f | band %.2f us (min %.2f max %.2f) | end %.2f..%.2f | cycles/wave: residual row %.0f, pass 1 %.0f, pass 2 %.0f, B+C+D %.0f (sum %.0f = %.2f us at 2.4 GHz)" % (
        k, m.sum(), st[m].min(), st[m].max(), band[m].mean(), band[m].min(), band[m].max(), end[m].min(), end[m].max(),
        t_z[m].mean(), t_p1[m].mean(), t_p2[m].mean(), t_bcd[m].mean(), tot.mean(), tot.mean() / 2400))
    print("   band-time histogram (us):", np.histogram(band[m], bins=[16, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 45, 50, 60])[0])
fastq = band < np.median(band)
for name, m in (("older half (band < median)", fastq & (dm == 0)), ("younger half", ~fastq & (dm == 0))):
    print("%s: n=%d band %.2f us | residual row %.0f pass 1 %.0f pass 2 %.0f B+C+D %.0f cycles" % (name, m.sum(), band[m].mean(), t_z[m].mean(), t_p1[m].mean(), t_p2[m].mean(), t_bcd[m].mean()))
simd = xcc * 100000 + ((hw >> 4) & 0xffff)  # everything above the wave slot
u, cnt = np.unique(simd, return_counts=True)
print("distinct SIMDs %d, waves per SIMD: min %d max %d" % (len(u), cnt.min(), cnt.max()))
