"""Persistent tile kernel against the workgroup-tile kernel, the direct kernels and (small sizes) the CPU oracle, plus
timing of both families on the cfg2 geometry:   python tools/persist_check.py [--hr 2048] [--no-oracle] [--time]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "super-resolution_amd", "python")):
    sys.path.insert(0, p)
import srmap as sr


def relerr(a, ref):
    a, ref = np.asarray(a, dtype=float).ravel(), np.asarray(ref, dtype=float).ravel()
    return float(np.max(np.abs(a - ref) / np.maximum(1.0, np.abs(ref))))


def one(ctx, W, H, K, s, shifts, use_oracle, label, C=1, b=3):
    rng = np.random.default_rng(7)
    w, h = W // s, H // s
    lr = rng.random((K, C, h, w))
    x = np.round(rng.random((C, H, W)) * 256) / 256
    wts = 0.5 + rng.random((C, H, W))
    p = sr.Problem(ctx, W, H, C, K, s, shifts, b, 1.0, sr.F64)
    p.set_observations(lr)
    p.set_irls_weights(p.add_regularizer(sr.REG_BTV, 0.01, 3, 0.5), wts)
    res = {}
    for name, impl in (("tiled", sr.IMPL_TILED), ("persist", sr.IMPL_PERSIST), ("direct", sr.IMPL_DIRECT)):
        p.set_impl(impl)
        res[name] = p.eval(x)
    f0, g0 = res["direct"]
    line = "%s: " % label
    for name in ("tiled", "persist"):
        f, g = res[name]
        line += "%s cost %.2e grad %.2e | " % (name, abs(f - f0) / max(1, abs(f0)), relerr(g, g0))
    if use_oracle:
        import oracle as orc
        model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=b, blur_sigma=1.0)
        ref = orc.Problem(model, lr)
        ref.add_regularizer(2, 0.01, 3, 0.5)
        ref.set_irls_weights(0, wts)
        fr, gr = ref.objective(x)
        f, g = res["persist"]
        line += "persist vs ORACLE cost %.2e grad %.2e" % (abs(f - fr) / max(1, abs(fr)), relerr(g, gr))
        bad = np.argwhere(np.abs(g - gr) / np.maximum(1, np.abs(gr)) > 1e-10)
        if len(bad):
            line += "  BAD px %d first %s rows %s..%s cols %s..%s" % (len(bad), bad[0], bad[:, 1].min(), bad[:, 1].max(), bad[:, 2].min(), bad[:, 2].max())
    else:
        f, g = res["persist"]
        bad = np.argwhere(np.abs(g - g0) / np.maximum(1, np.abs(g0)) > 1e-10)
        if len(bad):
            line += "  BAD px %d first %s rows %s..%s cols %s..%s" % (len(bad), bad[0], bad[:, 1].min(), bad[:, 1].max(), bad[:, 2].min(), bad[:, 2].max())
    print(line, flush=True)


def timing(ctx, W):
    import torch
    s, K = 4, 16
    shifts = [[k % s, (k // s) % s] for k in range(K)]
    p = sr.Problem(ctx, W, W, 1, K, s, shifts, 3, 1.0, sr.F64)
    y = torch.rand((K, 1, W // s, W // s), dtype=torch.float64, device="cuda")
    x = torch.rand((1, W, W), dtype=torch.float64, device="cuda"); g = torch.empty_like(x)
    p.set_observations_device(y.data_ptr())
    r = p.add_regularizer(sr.REG_BTV, 0.01, 3, 0.5)
    p.update_irls_weights_device(r, x.data_ptr())
    for name, impl in (("tiled", sr.IMPL_TILED), ("persist", sr.IMPL_PERSIST)):
        p.set_impl(impl)
        for _ in range(1000): p.eval_device(x.data_ptr(), g.data_ptr(), sr.TERM_ALL)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter(); n = 1000
            for _ in range(n): p.eval_device(x.data_ptr(), g.data_ptr(), sr.TERM_ALL)
            torch.cuda.synchronize()
            print("%d^2 f64 %s: %.2f us / evaluation" % (W, name, 1e6 * (time.perf_counter() - t0) / n), flush=True)


if __name__ == "__main__":
    import torch
    torch.cuda.init(); torch.zeros(1, device="cuda")
    ctx = sr.Context(0)
    s, K = 4, 16
    shifts = [[k % s, (k // s) % s] for k in range(K)]
    use_oracle = "--no-oracle" not in sys.argv
    one(ctx, 512, 512, K, s, shifts, use_oracle, "512x512 cfg2 shifts")
    one(ctx, 280, 92, K, s, shifts, use_oracle, "280x92 ragged")
    one(ctx, 1024, 64, K, s, [[(k * 3) % 7 - 3, (k * 5) % 7 - 3] for k in range(K)], use_oracle, "1024x64 +/-3 shifts")
    one(ctx, 2048, 2048, K, s, shifts, False, "2048x2048 cfg2")
    if "--time" in sys.argv:
        timing(ctx, 2048)
