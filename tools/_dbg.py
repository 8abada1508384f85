import os, sys, numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
sys.path.insert(0, "super-resolution_amd/python")
import srmap
W = 2048; s = 4; K = 16
shifts = [[k % s, (k // s) % s] for k in range(K)]
ctx = srmap.Context(0)
p = srmap.Problem(ctx, W, W, 1, K, s, shifts, 3, 1.0, srmap.F64); p.set_impl(srmap.IMPL_MARCH)
y = torch.rand((K, 1, W // s, W // s), dtype=torch.float64, device="cuda"); x = torch.rand((1, W, W), dtype=torch.float64, device="cuda")
p.set_observations_device(y.data_ptr()); r = p.add_regularizer(srmap.REG_BTV, 0.01, 3, 0.5); p.update_irls_weights_device(r, x.data_ptr())
g = torch.full_like(x, -7.0)
p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL); torch.cuda.synchronize()
g = g.cpu().numpy()[0]
nl = 512 * 512
for row in list(range(0, 8)) + [62, 63, 64, 65, 2040, 2046, 2047]:
    v = g[row, 0:4]; v2 = g[row, 256:260]
    print(row, [(int(a // nl), int(a % nl) // 512, int(a % 512), a % 1) for a in v], [(int(a // nl), int(a % nl) // 512, int(a % 512)) for a in v2])
gi = np.floor(g)
print("min", gi.min(), "max", gi.max(), "limit", 16 * nl)
bad = np.argwhere((gi < 0) | (gi >= 16 * nl))
print("bad count", len(bad), bad[:10])
ok = (g % 1) == 0
rows_bad = np.unique(np.argwhere(~ok)[:, 0]); print("rows flagged invalid:", rows_bad[:50])
