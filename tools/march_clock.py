"""Phase clock of the marching kernel (TIMING-ONLY build with -DSRMAP_EXP_MCLOCK=1, tools/exp_build_m.sh): per wave and
step the s_memtime stamps at the phase boundaries, read back out of g.   SRMAP_LIB=... python tools/march_clock.py"""
import os, sys, numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))
import srmap
W = 2048; s = 4; K = 16
shifts = [[k % s, (k // s) % s] for k in range(K)]
ctx = srmap.Context(0)
f32 = "--dtype" in sys.argv and sys.argv[sys.argv.index("--dtype") + 1] == "f32"
td = torch.float32 if f32 else torch.float64
SR = int(sys.argv[sys.argv.index("--sr") + 1]) if "--sr" in sys.argv else 16
p = srmap.Problem(ctx, W, W, 1, K, s, shifts, 3, 1.0, srmap.F32 if f32 else srmap.F64); p.set_impl(srmap.IMPL_MARCH)
y = torch.rand((K, 1, W // s, W // s), dtype=td, device="cuda"); x = torch.rand((1, W, W), dtype=td, device="cuda")
p.set_observations_device(y.data_ptr()); r = p.add_regularizer(srmap.REG_BTV, 0.01, 3, 0.5); p.update_irls_weights_device(r, x.data_ptr())
g = torch.empty_like(x)
for _ in range(300): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
torch.cuda.synchronize()
G = g.cpu().double().numpy()[0]
names = ["head (requests issued)", "phase 1", "wait vmcnt", "barrier 1", "phase 2 (to the store)", "store + barrier 2 + loop edge"]
rows = np.arange(W)
for strips, label in (([1, 2, 3, 4, 5, 6], "interior strips"), ([0], "first strip"), ([7], "last strip")):
    T = []
    for st in strips:
        c0 = st * 256
        T.append(np.stack([G[:, c0 + 0], G[:, c0 + 1], G[:, c0 + 2], G[:, c0 + 3], G[:, c0 + 4], G[:, c0 + 5]], axis=1))
    T = np.stack(T)                                  # [strip, row, stamp]
    T = T.reshape(len(strips), W // 64, 4, 16, 6)      # [strip, band, step, wave, stamp]
    d = np.diff(T, axis=-1)                            # phase durations within a step
    nxt = T[:, :, 1:, :, 0] - T[:, :, :-1, :, 5]       # store + barrier 2 + loop edge
    print("== %s: cycles per wave and step (mean over bands / steps; min .. max over waves)" % label)
    for k in range(5):
        m = d[..., k].mean(axis=(0, 1, 2))
        print("  %-30s %7.0f   (%5.0f .. %5.0f)" % (names[k], m.mean(), m.min(), m.max()))
    m = nxt.mean(axis=(0, 1, 2)); print("  %-30s %7.0f   (%5.0f .. %5.0f)" % (names[5], m.mean(), m.min(), m.max()))
    step = (T[:, :, 1:, :, 0] - T[:, :, :-1, :, 0]).mean()
    print("  step period %.0f cycles; band start -> last stamp %.0f cycles" % (step, (T[..., 5].max(axis=(2, 3)) - T[..., 0].min(axis=(2, 3))).mean()))
    print("  per-wave phase 1:", " ".join("%5.0f" % v for v in d[..., 1].mean(axis=(0, 1, 2))))
    print("  per-wave phase 2:", " ".join("%5.0f" % v for v in d[..., 4].mean(axis=(0, 1, 2))))

# stamps outside the loop (lane 2 / 3 of every wave, first step's rows): kernel entry, fill requested, fill landed + barrier,
# virtual step done, loop done
O = []
for st in range(8):
    c0 = st * 256 + 8
    O.append(np.stack([G[:, c0 + k] for k in range(5)], axis=1).reshape(W // 64, 64, 5)[:, :16, :])
O = np.stack(O)   # [strip, band, wave, stamp]
t0 = 0.0   # stamps are relative to each wave's own kernel entry
print("== outside the loop (cycles, mean over workgroups and waves; relative to the chip's first kernel-entry stamp)")
for k, nm in enumerate(["kernel entry", "window requested", "window landed + barrier", "virtual step + inputs of step 0 + barrier", "loop done"]):
    print("  %-44s %8.0f  (min %7.0f max %7.0f)" % (nm, (O[..., k] - t0).mean(), (O[..., k] - t0).min(), (O[..., k] - t0).max()))
