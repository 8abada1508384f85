"""One-GPU evaluation time of the five BASELINE.json configurations (device-generated data, f64 unless noted):
   python tools/config_timing.py [cfg ...]      e.g. 1 2 3 4 5"""
import os, sys, time
import numpy as np, torch
torch.cuda.init(); torch.zeros(1, device='cuda')  # torch's HIP runtime before libsrmap.so
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))
import srmap

CFG = {
    1: dict(name="cfg1 4 frames, 2x -> 256^2, TV", W=256, C=1, K=4, s=2, blur=(0, 0.0), regs=[(srmap.REG_TV, 0, 0.0)]),
    2: dict(name="cfg2 16 frames, 4x -> 2048^2, blur + BTV", W=2048, C=1, K=16, s=4, blur=(3, 1.0), regs=[(srmap.REG_BTV, 3, 0.5)]),
    3: dict(name="cfg3 16 frames RGB, 4x -> 4096^2, BTV", W=4096, C=3, K=16, s=4, blur=(0, 0.0), regs=[(srmap.REG_BTV, 3, 0.5)]),
    4: dict(name="cfg4 9 frames x 128 ch, 3x -> 1023^2, TV", W=1023, C=128, K=9, s=3, blur=(0, 0.0), regs=[(srmap.REG_TV, 0, 0.0)]),
    5: dict(name="cfg5 64 frames x 256 ch, 4x -> 2048^2, BTV + 3-D TV", W=2048, C=256, K=64, s=4, blur=(0, 0.0),
            regs=[(srmap.REG_BTV, 3, 0.5), (srmap.REG_TV3D, 0, 0.0)]),
}


def main():
    blurred = "--blur" in sys.argv  # also time cfg3-5 with the Gaussian blur (3, 1.0) cfg2 states
    sys.argv = [a for a in sys.argv if a != "--blur"]
    if blurred:
        for c in (3, 4, 5):
            CFG[c]["blur"] = (3, 1.0); CFG[c]["name"] += " + blur"
    chan = None
    if "--channels" in sys.argv:  # override the channel count (per-kernel profiles of cfg4 / cfg5 at a few channels)
        i = sys.argv.index("--channels"); chan = int(sys.argv[i + 1]); del sys.argv[i:i + 2]
    which = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
    dev = torch.device("cuda", 0)
    ctx = srmap.Context(0)
    for c in which:
        cf = CFG[c]
        W, C, K, s = cf["W"], chan or cf["C"], cf["K"], cf["s"]
        w = W // s
        shifts = [[k % s, (k // s) % s] for k in range(K)]
        g = torch.Generator(device=dev); g.manual_seed(c)
        y = torch.rand((K, C, w, w), generator=g, device=dev, dtype=torch.float64)
        x = torch.rand((C, W, W), generator=g, device=dev, dtype=torch.float64)
        gr = torch.empty_like(x)
        p = srmap.Problem(ctx, W, W, C, K, s, shifts, cf["blur"][0], cf["blur"][1], srmap.F64)
        p.set_observations_device(y.data_ptr())
        for kind, r, d in cf["regs"]:
            ri = p.add_regularizer(kind, 0.01, r, d)
            p.update_irls_weights_device(ri, x.data_ptr())  # IRLS weights resident, as in the solver loop
        n = 3 if C >= 128 else (30 if C * W * W >= 4096 * 4096 else 1000)
        for _ in range(2):
            p.eval_device(x.data_ptr(), gr.data_ptr(), srmap.TERM_ALL)
        torch.cuda.synchronize()
        tr = time.perf_counter()  # sustained clocks: at least 100 ms of load before the timed evaluations (as bench.py)
        while time.perf_counter() - tr < 0.1:
            for _ in range(10):
                p.eval_device(x.data_ptr(), gr.data_ptr(), srmap.TERM_ALL)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            p.eval_device(x.data_ptr(), gr.data_ptr(), srmap.TERM_ALL)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        N, nn = W * W, w * w
        balg = 8 * C * ((2 + len(cf["regs"])) * N + K * nn)
        print("%-52s %10.3f ms / evaluation  %7.1f us / channel  B_alg %.2f GB -> %.0f GB/s (%.1f%% of 8 TB/s)" %
              (cf["name"], dt * 1e3, dt * 1e6 / C, balg / 1e9, balg / dt / 1e9, 100 * balg / dt / 8e12))
        del p, x, y, gr
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
