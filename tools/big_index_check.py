"""64-bit indexing check at cfg5-like extents (K * C * n > 2^31 elements): the gradient of the LAST
channels of a many-channel problem must equal that of a single-channel problem built from the same slices.
Everything is generated on the device (no multi-GB host arrays).  python tools/big_index_check.py [C] [K]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))
import srmap


def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 136
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    s, W = 4, 2048
    H, w, h = W, W // s, W // s
    dev = torch.device("cuda", 0)
    shifts = [[k % s, (k // s) % s] for k in range(K)]
    ctx = srmap.Context(0)
    big = srmap.Problem(ctx, W, H, C, K, s, shifts, 3, 1.0, srmap.F32)
    g = torch.Generator(device=dev); g.manual_seed(1)
    y = torch.rand((K, C, h, w), generator=g, device=dev, dtype=torch.float32)
    print("observations: %.2f G elements (2^31 = 2.147 G)" % (y.numel() / 1e9))
    big.set_observations_device(y.data_ptr())
    big.add_regularizer(srmap.REG_BTV, 0.01, 3, 0.5)
    x = torch.rand((C, H, W), generator=g, device=dev, dtype=torch.float32)
    gbig = torch.empty_like(x)
    cost_big = big.eval_device(x.data_ptr(), gbig.data_ptr(), srmap.TERM_ALL, want_cost=True)
    torch.cuda.synchronize()
    ok = True
    for c in (0, C // 2, C - 1):
        one = srmap.Problem(ctx, W, H, 1, K, s, shifts, 3, 1.0, srmap.F32)
        yc = y[:, c:c + 1].contiguous()
        one.set_observations_device(yc.data_ptr())
        one.add_regularizer(srmap.REG_BTV, 0.01, 3, 0.5)
        xc = x[c:c + 1].contiguous()
        gc = torch.empty_like(xc)
        one.eval_device(xc.data_ptr(), gc.data_ptr(), srmap.TERM_ALL, want_cost=True)
        torch.cuda.synchronize()
        err = float((gc[0] - gbig[c]).abs().max() / gc.abs().max())
        print("channel %d: max relative gradient difference %.3e" % (c, err))
        ok = ok and err == 0.0
        del one
    print("cost", cost_big, "finite", np.isfinite(cost_big))
    print("BIG INDEX CHECK", "PASSED" if ok and np.isfinite(cost_big) else "FAILED")


if __name__ == "__main__":
    main()
