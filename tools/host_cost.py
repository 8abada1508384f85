import os, sys, time
import numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
sys.path.insert(0, "super-resolution_amd/python")
import srmap
ctx = srmap.Context(0)
for W in (64, 2048):
    s, K = 4, 16
    shifts = [[k % s, (k // s) % s] for k in range(K)]
    p = srmap.Problem(ctx, W, W, 1, K, s, shifts, 3, 1.0, srmap.F64)
    y = torch.rand((K, 1, W // s, W // s), dtype=torch.float64, device="cuda")
    p.set_observations_device(y.data_ptr())
    p.add_regularizer(srmap.REG_BTV, 0.01, 3, 0.5)
    x = torch.rand((1, W, W), dtype=torch.float64, device="cuda"); g = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(20): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL, stream=st)
    torch.cuda.synchronize()
    n = 500
    t0 = time.perf_counter()
    for _ in range(n): p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL, stream=st)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("W=%d host enqueue per eval %.1f us, total per eval %.1f us" % (W, 1e6*(t1-t0)/n, 1e6*(t2-t0)/n))
    # sync each time
    t0 = time.perf_counter()
    for _ in range(100):
        p.eval_device(x.data_ptr(), g.data_ptr(), srmap.TERM_ALL, stream=st); torch.cuda.synchronize()
    print("   eval + sync each: %.1f us" % (1e6*(time.perf_counter()-t0)/100))
