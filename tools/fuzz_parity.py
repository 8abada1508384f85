"""Randomised parity sweep: fused (AUTO) path vs the CPU oracle on random geometries, shifts and regulariser
mixes (f64: 1e-11 relative on the gradient and the cost; f32: 1e-4 / 2e-5).
   python tools/fuzz_parity.py [cases] [seed] [f64|f32]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "super-resolution_amd", "python")):
    sys.path.insert(0, p)
import oracle as orc
import srmap


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    f32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
    tol_g, tol_f = (1e-4, 2e-5) if f32 else (1e-11, 1e-11)
    rng = np.random.default_rng(seed)
    ctx = srmap.Context(0)
    worst = 0.0
    tiled = 0
    for it in range(cases):
        s = int(rng.integers(2, 5))
        h, w = int(rng.integers(3, 40)), int(rng.integers(3, 90))
        H, W = h * s, w * s
        C = int(rng.integers(1, 4))
        K = int(rng.integers(1, 21))
        span = int(rng.integers(0, 7))
        shifts = [[int(v) for v in rng.integers(-span, span + 1, 2)] for _ in range(K)]
        b = int(rng.choice([0, 3]))
        sigma = float(rng.uniform(0.6, 1.6)) if b else 0.0
        regs = []
        for _ in range(int(rng.integers(0, 3))):
            kind = int(rng.choice([srmap.REG_TV, srmap.REG_TV3D, srmap.REG_BTV]))
            regs.append((kind, float(rng.uniform(0.005, 0.05)), int(rng.integers(1, 4)), float(rng.uniform(0.3, 1.0))))
        model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=b, blur_sigma=sigma)
        lr = rng.random((K, C, h, w))
        ref = orc.Problem(model, lr)
        p = srmap.Problem(ctx, W, H, C, K, s, shifts, b, sigma, srmap.F32 if f32 else srmap.F64)
        p.set_observations(lr)
        for kind, lam, rg, dc in regs:
            i = p.add_regularizer(kind, lam, rg, dc)
            ref.add_regularizer(kind, lam, rg, dc)
            wts = 0.5 + 2 * rng.random((C, H, W))
            p.set_irls_weights(i, wts); ref.set_irls_weights(i, wts)
        x = np.round(rng.random((C, H, W)) * 32) / 32   # exact ties exercise sgn(0)
        f_ref, g_ref = ref.objective(x)
        f, g = p.eval(x)
        eg = float(np.max(np.abs(np.ravel(g) - np.ravel(g_ref)) / np.maximum(1.0, np.abs(np.ravel(g_ref)))))
        ef = abs(f - f_ref) / max(1.0, abs(f_ref))
        worst = max(worst, eg, ef)
        try:
            p.set_impl(srmap.IMPL_TILED); p.eval(x); tiled += 1
        except srmap.SrmapError:
            pass
        if eg > tol_g or ef > tol_f:
            print("MISMATCH case", it, dict(s=s, W=W, H=H, C=C, K=K, shifts=shifts, b=b, regs=regs), eg, ef)
            sys.exit(1)
    print("fuzz ok: %d cases (%d on the fused path), worst relative error %.2e" % (cases, tiled, worst))


if __name__ == "__main__":
    main()
