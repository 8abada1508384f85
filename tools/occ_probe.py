"""Occupancy probe:  python tools/occ_probe.py
Times the tile kernel per megapixel for geometries whose instances are compiled for different occupancies (all f64, blur
3 / 1.0, BTV(3, .5), K = S * S frames on distinct phases, 2048^2 HR):  S = 4 (4 waves / SIMD), S = 2 (6 waves / SIMD),
S = 3 (4 waves / SIMD).  us per evaluation in a tight loop after a clock ramp, best of 4 x 1000."""
import os, sys, time
import torch
torch.cuda.init(); torch.zeros(1, device="cuda")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))
import srmap

def run(W, s, K, dtype, reg=(srmap.REG_BTV, 3, 0.5), blur=3):
    tdt = torch.float64 if dtype == srmap.F64 else torch.float32
    shifts = [[k % s, (k // s) % s] for k in range(K)]
    ctx = srmap.Context(0)
    p = srmap.Problem(ctx, W, W, 1, K, s, shifts, blur, 1.0, dtype)
    gen = torch.Generator(device="cuda"); gen.manual_seed(7)
    y = torch.rand((K, 1, W // s, W // s), dtype=tdt, device="cuda", generator=gen)
    x = torch.rand((1, W, W), dtype=tdt, device="cuda", generator=gen); g = torch.empty_like(x)
    p.set_observations_device(y.data_ptr())
    r = p.add_regularizer(reg[0], 0.01, reg[1], reg[2])
    p.update_irls_weights_device(r, x.data_ptr())
    for _ in range(2000): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 1000
        for _ in range(n): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL)
        torch.cuda.synchronize(); best = min(best, 1e6 * (time.perf_counter() - t0) / n)
    return best

if __name__ == "__main__":
    # python tools/occ_probe.py [f64|f32] [cfg2]   (SRMAP_LIB selects a measurement build)
    only = [a for a in sys.argv[1:] if a in ("f64", "f32")]
    geos = ((2048, 4, 16),) if "cfg2" in sys.argv else ((2048, 4, 16), (2048, 2, 4), (2046, 3, 9), (2048, 2, 16), (4096, 4, 16), (4096, 2, 4))
    for (W, s, K) in geos:
        for dt, nm in ((srmap.F64, "f64"), (srmap.F32, "f32")):
            if only and nm not in only: continue
            t = run(W, s, K, dt)
            print("W %d S %d K %2d %s  %7.2f us  %6.2f us/Mpx" % (W, s, K, nm, t, t / (W * W / 1e6)), flush=True)
