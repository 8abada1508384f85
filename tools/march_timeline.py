"""Development build only (SRMAP_DEV_INSTANCES): per-wave time stamps of the marching kernel on cfg2.
   python tools/march_timeline.py"""
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
st = (d[:, 0] - t0) * tick; twait = d[:, 1].astype(np.float64); pro = (d[:, 2] - d[:, 0]) * tick; fill = pro * 0
loop = (d[:, 3] - d[:, 2]) * tick; tail = (d[:, 4] - d[:, 3]) * tick; end = (d[:, 4] - t0) * tick
slow = (d[:, 5] & 1) == 1; duty = (d[:, 5] & 2) == 2; ring = (d[:, 5] & 4) == 4
print("cycles waiting at B per wave: mean %.0f min %.0f max %.0f (of ~%.0f cycles of band time at 2.4 GHz)" % (twait.mean(), twait.min(), twait.max(), loop.mean() * 2400))
print("waves %d, kernel span %.2f us (first start -> last end)" % (n, end.max()))
def desc(name, m):
    if m.sum() == 0: return
    print("%-14s n=%4d start %.2f..%.2f | prologue(issue+duty+halo) %.2f (max %.2f) | ring fill wait %.2f | loop %.2f (min %.2f max %.2f) | ticket %.2f | end %.2f..%.2f" % (
        name, m.sum(), st[m].min(), st[m].max(), pro[m].mean(), pro[m].max(), fill[m].mean(), loop[m].mean(), loop[m].min(), loop[m].max(), tail[m].mean(), end[m].min(), end[m].max()))
desc("all", np.ones(n, bool))
desc("fast", ~slow)
desc("slow", slow)
desc("duty", duty)
desc("fast non-duty", ~slow & ~duty)
desc("slow non-duty", slow & ~duty)
desc("ring waves", ring)
dm = ((d[:, 5] >> np.uint64(4)) & np.uint64(3)).astype(int)
for k in range(3): desc("dm=%d" % k, dm == k)
strip = (d[:, 6] >> np.uint64(32)).astype(int); band = (d[:, 6] & np.uint64(0xffffffff)).astype(int)
for sidx in range(strip.max() + 1):
    m = strip == sidx
    print("strip %d: n=%d loop mean %.2f min %.2f max %.2f end max %.2f" % (sidx, m.sum(), loop[m].mean(), loop[m].min(), loop[m].max(), end[m].max()))
print("dm=0 band-time histogram (us):", np.histogram(loop[dm == 0], bins=[20,22,24,26,28,30,32,34,36,38,40,45,50,60])[0])
print("dm=1 band-time histogram (us):", np.histogram(loop[dm == 1], bins=[20,22,24,26,28,30,32,34,36,38,40,45,50,60])[0])
hwk = xcc * 10000 + ((d[:, 7] & np.uint64(0xffffffff)).astype(np.int64) >> 4 & 0xfff)
import collections
grp = collections.defaultdict(list)
for i in range(n): grp[int(hwk[i])].append(i)
pairs = [(loop[v[0]], loop[v[1]], dm[v[0]], dm[v[1]]) for v in grp.values() if len(v) == 2]
pa = np.array(pairs)
print("SIMD pairs: %d; corr of band times within a pair %.2f; mean |diff| %.2f" % (len(pa), np.corrcoef(pa[:,0], pa[:,1])[0,1], np.abs(pa[:,0]-pa[:,1]).mean()))
print("pair max-time histogram:", np.histogram(np.maximum(pa[:,0],pa[:,1]), bins=[20,24,28,32,36,40,45,50,60])[0])
late = np.argsort(-end)[:12]
for i in late: print("late wave strip %d band %d dm %d xcc %d start %.2f loop %.2f end %.2f" % (strip[i], band[i], dm[i], int(d[i,7] >> np.uint64(32)), st[i], loop[i], end[i]))
xcc = (d[:, 7] >> np.uint64(32)).astype(int)
for k in range(8):
    m = xcc == k
    print("xcc %d: n=%d slow=%d duty=%d end max %.2f loop mean %.2f" % (k, m.sum(), (slow & m).sum(), (duty & m).sum(), end[m].max() if m.sum() else 0, loop[m].mean() if m.sum() else 0))
hw = (d[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
key = xcc * 10000 + se * 1000 + cu * 10 + simd
u, cnt = np.unique(key, return_counts=True)
print("distinct SIMDs %d, waves per SIMD: min %d max %d" % (len(u), cnt.min(), cnt.max()))
