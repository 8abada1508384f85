"""Chip-wide timeline of one cfg2 evaluation through the timing-only library (build.sh): when every tile workgroup ran
(s_memrealtime, 100 MHz, common to all XCDs) and on which CU (HW_ID / XCC_ID) -> slot occupancy, ramp and tail."""
import os, sys, ctypes
import numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))
import srmap
srmap.LIB_PATH = os.path.join(ROOT, "tools", "phase_clock", "libsrmap_time.so")
W = 2048; s, K = 4, 16
shifts = [[k % s, (k // s) % s] for k in range(K)]
ctx = srmap.Context(0)
p = srmap.Problem(ctx, W, W, 1, K, s, shifts, 3, 1.0, srmap.F64)
y = torch.rand((K, 1, W // s, W // s), dtype=torch.float64, device="cuda")
x = torch.rand((1, W, W), dtype=torch.float64, device="cuda"); g = torch.empty_like(x)
p.set_observations_device(y.data_ptr())
r = p.add_regularizer(srmap.REG_BTV, 0.01, 3, 0.5)
p.update_irls_weights_device(r, x.data_ptr())
for _ in range(3000): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
torch.cuda.synchronize()
nb = 8 * 256
buf = torch.zeros((nb + 512, 8, 16), dtype=torch.int64, device="cuda")
L = srmap.load()
L.srmap_dbg_set_timing.argtypes = [ctypes.c_void_p]
assert L.srmap_dbg_set_timing(ctypes.c_void_p(buf.data_ptr())) == 0
for _ in range(5): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
torch.cuda.synchronize()
L.srmap_dbg_set_timing(ctypes.c_void_p(0))
tall = buf.cpu().numpy()
t = tall[:nb]
bb = tall[nb:, 0, :]
bb = bb[bb[:, 10] > 0]
t0 = t[:, :, 10].min(axis=1).astype(np.float64); t1 = t[:, :, 11].max(axis=1).astype(np.float64)  # per workgroup, 10 ns ticks
hw = t[:, 0, 12]
xcc = (hw >> 32) & 0xf; hwid = hw & 0xffffffff
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
base = t0.min(); t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0  # us
if len(bb):
    real = bb[bb[:, 13] == 1]
    print("border blocks: %d (%d with work); start %.2f..%.2f us, end %.2f..%.2f us, life of the working ones mean %.2f max %.2f us" % (
        len(bb), len(real), (bb[:, 10].min() - base) / 100.0, (bb[:, 10].max() - base) / 100.0, (bb[:, 11].min() - base) / 100.0,
        (bb[:, 11].max() - base) / 100.0, ((real[:, 11] - real[:, 10]) / 100.0).mean(), ((real[:, 11] - real[:, 10]) / 100.0).max()))
print("tile workgroups: first start 0, last start %.2f us, first end %.2f, last end %.2f us" % (t0.max(), t1.min(), t1.max()))
print("workgroup life: mean %.2f us  min %.2f  max %.2f" % ((t1 - t0).mean(), (t1 - t0).min(), (t1 - t0).max()))
cus = np.unique(key)
print("distinct CUs seen: %d; tiles per CU: min %d max %d" % (len(cus), min((key == c).sum() for c in cus), max((key == c).sum() for c in cus)))
by = np.arange(nb) // 256
for b in range(8):
    m = by == b
    print("tile column %d: start %.2f..%.2f  end %.2f..%.2f  life %.2f" % (b, t0[m].min(), t0[m].max(), t1[m].min(), t1[m].max(), (t1 - t0)[m].mean()))
# occupancy over time: resident tile workgroups per 1-us bin
T = int(np.ceil(t1.max())) + 1
occ = np.zeros(T)
for a, b in zip(t0, t1):
    i0, i1 = int(a), int(b)
    for i in range(i0, min(i1 + 1, T)):
        occ[i] += min(b, i + 1) - max(a, i)
print("resident tile workgroups by microsecond (of 512 slots):")
print(" ".join("%3.0f" % v for v in occ))
busy = np.array([sum((t1 - t0)[key == c]) for c in cus])
print("per-CU sum of workgroup lives: mean %.1f us  min %.1f  max %.1f  (kernel span %.1f us, 2 slots per CU)" % (busy.mean(), busy.min(), busy.max(), t1.max()))
last = np.array([t1[key == c].max() for c in cus]); first = np.array([t0[key == c].min() for c in cus])
print("per-CU last end: min %.2f  mean %.2f  max %.2f;  first start: min %.2f mean %.2f max %.2f" % (last.min(), last.mean(), last.max(), first.min(), first.mean(), first.max()))
