#!/usr/bin/env python3
"""bench.py -- MAP gradient iterations/s of the MI355X-native path.

One "step" = one MAP gradient iteration = one ObjectiveFunction::ComputeAllTerms
(data term over all K frames + IRLS-weighted regulariser, cost and gradient,
reference src/optimization/objective_function.cpp:5-20) on device-resident
synthetic inputs.

N = 1 workload: BASELINE.json configs[1] -- 16-frame grayscale, 4x upscale to
2048x2048 HR, Gaussian blur (3, sigma 1) + BTV (range 3, decay 0.5, lambda 0.01).
N > 1 (one process per GPU, torch.distributed over RCCL): the path shards by
channel (the reference's split_channels semantics, irls_map_solver.cpp:200-262):
rank r owns channel r of an N-channel problem of the same per-channel geometry,
so per-GPU work is fixed ("weak" scaling) and -- these being the reference's
independent per-channel solves -- there is no collective in the timed region.
`value` counts channel-iterations per second summed over ranks (at N = 1 this is
plain iterations per second).  `--shard frames` runs the north-star's
frame-sharded variant instead (each rank K/N frames of the SAME image + RCCL
all-reduce of the HR gradient), `--shard rows` the row-band variant (halo rows of
x exchanged with ncclSend / ncclRecv); both go through the library's own
sharded evaluation (srmap_eval_sharded_device: the exchange is issued by the C
ABI on the evaluation's stream) and are reported as strong scaling.

Timing: W untimed warm-up steps, then exactly K timed steps between barriers
(defaults K = 2000, W = 200).  Before the warm-up the GPU is driven for
--clock-ramp-ms (default 100 ms) of untimed evaluations: an MI355X reaches its
sustained clocks only after ~50 ms of load, and a 0.06 ms step measured in the
first few hundred launches reads ~10 % slower than the same step in a running
solver (the count is reported as config.clock_ramp_steps_before_warmup).

Prints ONE JSON line on rank 0 (see the task contract), including
  "roofline":     algorithmic bytes of one step / mean step time vs 8 TB/s HBM
  "cpu_baseline": the CPU oracle (a port of the reference, oracle/) timed on
                  this host on a bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def synth_ground_truth(W, H, C):
    """smooth + edges, channel scaled."""
    u, v = np.meshgrid((np.arange(W) + 0.5) / W, (np.arange(H) + 0.5) / H)
    base = 0.5 + 0.25 * np.sin(2 * np.pi * 3 * u) * np.cos(2 * np.pi * 5 * v) \
        + 0.25 * (((u - .5) ** 2 + (v - .5) ** 2) < .09)
    base = np.clip(base, 0, 1)
    scale = [(0.6 + 0.4 * c / (C - 1)) if C > 1 else 1.0 for c in range(C)]
    return np.stack([base * s for s in scale])


def bilinear_upsample(img, s):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(img))[None]
    return torch.nn.functional.interpolate(t, scale_factor=s, mode="bilinear", align_corners=False)[0].numpy()


def pmc_traffic(dtype):
    """HBM bytes per launch of the fused kernel from the committed PMC profile
    of this same command (bench.py cannot run rocprofv3 on itself)."""
    path = os.path.join(ROOT, "profiles", "r02_bench_hbm_pmc.json")
    try:
        with open(path) as f:
            rec = json.load(f)[dtype]
        return (2.0 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024.0
    except Exception:
        return None


def cpu_baseline(cfg, lr, x0, wts, budget_s=12.0):
    """Oracle (CPU restatement of the reference, 1 thread like the reference) on
    a bounded crop of the same workload, scaled by pixel count."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    s, K = cfg["scale"], cfg["frames"]
    crop_lr = 256  # LR crop 256x256 -> HR 1024x1024: 1/4 of the cfg2 pixels (~0.65 s per evaluation)
    crop_lr = min(crop_lr, lr.shape[-1])
    ch = crop_lr * s
    model = orc.ImageModel(scale=s, shifts=cfg["shifts"], blur_ksize=cfg["blur"][0], blur_sigma=cfg["blur"][1])
    prob = orc.Problem(model, lr[:, :1, :crop_lr, :crop_lr])
    prob.add_regularizer(orc.REG_BTV, cfg["lambda"], cfg["btv"][0], cfg["btv"][1])
    prob.set_irls_weights(0, wts[:1, :ch, :ch])
    x = np.ascontiguousarray(x0[:1, :ch, :ch])
    t0 = time.perf_counter()
    n = 0
    while True:
        prob.objective(x)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 40:  # about 10-15 s of CPU work
            break
    per_eval_crop = el / n
    frac = (ch * ch) / float(cfg["W"] * cfg["H"])
    per_eval_full = per_eval_crop / frac
    return {"value": 1.0 / per_eval_full, "unit": "MAP gradient iterations/s", "cores": 1, "kind": "port",
            "sample": "%d evaluations of a %dx%d HR crop (%d frames, same blur/BTV), %.2f s each, scaled by "
                      "pixel count x%.0f to the full %dx%d" % (n, ch, ch, K, per_eval_crop, 1 / frac, cfg["W"], cfg["H"]),
            "ms_per_step": per_eval_full * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--clock-ramp-ms", type=float, default=100.0,
                    help="untimed evaluations for this many milliseconds BEFORE the W warm-up steps: the GPU reaches its "
                         "sustained clocks only after ~50 ms of load (a 0.06 ms step measured cold reads 10 %% slower than "
                         "the same step 1000 steps later); 0 disables")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64",
                    help="arithmetic/storage type on device (the reference is f64)")
    ap.add_argument("--shard", choices=["channels", "frames", "rows"], default="channels",
                    help="N > 1: channels = one cfg2 channel per GPU, no collective (weak, default); frames = frame shards + "
                         "RCCL all-reduce of the gradient (strong); rows = HR row bands + halo exchange of x (strong)")
    ap.add_argument("--impl", choices=["auto", "direct", "tiled"], default="auto")
    ap.add_argument("--hr", type=int, default=2048)
    ap.add_argument("--terms", choices=["all", "data", "reg"], default="all",
                    help="ablation only: evaluate a subset of the objective terms")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--test-single-device", action="store_true",
                    help="testing aid for 1-GPU boxes: all ranks share GPU 0 and talk over gloo (exercises the N > 1 code "
                         "paths; the numbers mean nothing)")
    ap.add_argument("--joint-scalars", action="store_true",
                    help="channel sharding only: also all-reduce the scalar cost every step, as ONE joint solve over "
                         "all channels would (srmap_solve_sharded). Default: the reference's split_channels semantics "
                         "(irls_map_solver.cpp:200-210), independent per-channel solves, no collective in the timed region")
    args = ap.parse_args()

    import torch
    import srmap

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.test_single_device else int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.test_single_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    s, K = 4, 16
    W = H = args.hr
    w, h = W // s, H // s
    shifts_all = [[k % s, (k // s) % s] for k in range(K)]
    cfg = {"W": W, "H": H, "scale": s, "frames": K, "shifts": shifts_all, "blur": (3, 1.0),
           "btv": (3, 0.5), "lambda": 0.01}
    dtype = srmap.F64 if args.dtype == "f64" else srmap.F32
    tdtype = torch.float64 if args.dtype == "f64" else torch.float32
    E = 8 if args.dtype == "f64" else 4

    import srmap_dist
    frame_ids = list(range(K))
    units_per_step = 1.0
    if world > 1 and args.shard == "frames":
        frame_ids = srmap_dist.frame_shard(K, world, rank)
    shifts = [shifts_all[k] for k in frame_ids]
    Kloc = len(frame_ids)

    # row bands: this rank's problem lives on its owned HR rows + halo rows
    band = None
    Hloc, e0, e1, r0, r1 = H, 0, H, 0, H
    if world > 1 and args.shard == "rows":
        halo = srmap_dist.band_halo(s, 3, s - 1, 3)
        bands = [srmap_dist.row_band(H, s, world, r, halo) for r in range(world)]
        (r0, r1), (e0, e1) = bands[rank]
        Hloc = e1 - e0

    ctx = srmap.Context(local_rank)
    prob = srmap.Problem(ctx, W, Hloc, 1, Kloc, s, shifts, 3, 1.0, dtype)
    prob.set_impl({"auto": srmap.IMPL_AUTO, "direct": srmap.IMPL_DIRECT, "tiled": srmap.IMPL_TILED}[args.impl])

    # ---- synthetic data, seeded; channel = rank under channel sharding
    rng = np.random.default_rng(20240607 + (rank if args.shard == "channels" else 0))
    gt = synth_ground_truth(W, H, max(world, 1) if args.shard == "channels" else 1)
    gt = gt[rank:rank + 1] if args.shard == "channels" and world > 1 else gt[:1]
    gt = gt[:, e0:e1, :]
    lr = np.stack([prob.apply(gt, i) for i in range(Kloc)])
    noise_rng = np.random.default_rng(777 + rank)
    lr = lr + (5.0 / 255.0) * noise_rng.standard_normal(lr.shape)
    prob.set_observations(lr)
    reg = prob.add_regularizer(srmap.REG_BTV, cfg["lambda"], 3, 0.5)
    x0 = bilinear_upsample(lr[0], s)
    rv0 = prob.reg_values(reg, x0)
    wts = 1.0 / np.maximum(1e-5, rv0)
    prob.set_irls_weights(reg, wts)

    # ---- the library's communicator and shard description (N > 1, frames / rows): the exchanges run inside the C ABI
    comm, sd = None, None
    if world > 1 and args.shard in ("frames", "rows"):
        if args.test_single_device:
            comm = srmap.Comm(ctx, rank, world, backend="host", dist=dist)
        else:
            uid = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                uid = torch.frombuffer(bytearray(srmap.Comm.unique_id(ctx)), dtype=torch.uint8).to(dev)
            dist.broadcast(uid, 0)
            comm = srmap.Comm(ctx, rank, world, backend="rccl", unique_id=bytes(uid.cpu().numpy().tobytes()))
        sd = srmap.ShardDesc()
        if args.shard == "frames":
            sd.mode, sd.reg_rank = srmap.SHARD_FRAMES, 0
        else:
            prob.set_cost_rows(r0 - e0, r1 - e0)
            sd.mode = srmap.SHARD_ROWS
            sd.own_row0, sd.own_row1 = r0 - e0, r1 - e0
            if rank + 1 < world:
                (n0, n1), (ne0, ne1) = bands[rank + 1]
                sd.send_down_rows = n0 - ne0
            if rank > 0:
                (u0, u1), (ue0, ue1) = bands[rank - 1]
                sd.send_up_rows = ue1 - u1

    x_dev = torch.from_numpy(x0).to(dev, tdtype).contiguous()
    g_dev = torch.empty_like(x_dev)
    stream = torch.cuda.Stream(device=dev)
    sh = stream.cuda_stream
    cost_buf = torch.zeros(1, dtype=torch.float64, device=dev)
    terms = {"all": srmap.TERM_ALL, "data": srmap.TERM_DATA, "reg": srmap.TERM_REG}[args.terms]

    def step():
        if comm is not None:
            # halo rows of x (rows) / gradient + cost all-reduce (frames): issued by the library on `sh`
            prob.eval_sharded_device(comm, sd, x_dev.data_ptr(), g_dev.data_ptr(), terms, want_cost=False, stream=sh)
        else:
            prob.eval_device(x_dev.data_ptr(), g_dev.data_ptr(), terms, want_cost=False, stream=sh)
            if dist is not None and args.joint_scalars:
                with torch.cuda.stream(stream):
                    dist.all_reduce(cost_buf)

    def barrier():
        stream.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ramp_steps = 0
    if args.clock_ramp_ms > 0:  # untimed: bring the GPU to its sustained clock state (see --clock-ramp-ms)
        t_r = time.perf_counter()
        while (time.perf_counter() - t_r) * 1e3 < args.clock_ramp_ms:
            for _ in range(50):
                step()
            stream.synchronize()
            ramp_steps += 50
    for _ in range(args.warmup):
        step()
    barrier()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        ev0.record(stream)
    for _ in range(args.steps):
        step()
    with torch.cuda.stream(stream):
        ev1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)  # HIP events on the stream the kernels run on

    tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())
    ms_per_step = wall / args.steps * 1e3
    if world > 1 and args.shard == "channels":
        units_per_step = float(world)
    value = units_per_step * args.steps / wall

    if rank == 0:
        C_total = world if (world > 1 and args.shard == "channels") else 1
        N, n = W * H, w * h
        rho = 1
        b_alg = E * C_total * ((2 + rho) * N + K * n)  # x, y, w read once, g written once
        step_dev_s = dev_ms / args.steps * 1e-3
        achieved = (b_alg / max(C_total, 1)) / step_dev_s / 1e9  # per GPU
        out = {
            "metric": "MAP gradient iterations/sec at fixed HR size",
            "value": value,
            "unit": "MAP gradient iterations/s" if C_total == 1 else "channel-iterations/s (one 16-frame %dx%d channel per GPU)" % (W, H),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if (world > 1 and args.shard in ("frames", "rows")) else "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: 16-frame grayscale, 4x upscale to %dx%d, Gaussian blur 3/1.0 + BTV(3,0.5) "
                                   "lambda 0.01, IRLS weights from x0" % (W, H),
                       "frames": K, "scale": s, "channels": C_total, "shard": args.shard if world > 1 else "none",
                       "collective_per_step": ("none" if world == 1 else
                                               "ncclAllReduce(g, C*N) + ncclAllReduce(cost) inside srmap_eval_sharded_device" if args.shard == "frames" else
                                               "ncclSend/ncclRecv of the halo rows of x inside srmap_eval_sharded_device" if args.shard == "rows" else
                                               "all-reduce(cost)" if args.joint_scalars else
                                               "none (split_channels: independent per-channel solves)"),
                       "impl": args.impl, "device_ms_per_step": dev_ms / args.steps,
                       "clock_ramp_steps_before_warmup": ramp_steps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args.dtype),
                         "traffic_source": "profiles/r02_bench_hbm_pmc.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                           "(separate passes) of this command; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, "
                                           "the factor 2 being the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md",
                         "algorithmic_bytes_per_step": b_alg / max(C_total, 1),
                         "kernel": "whole evaluation (k_eval_z + k_finish_eval: all kernels of one step), HIP events on "
                                   "the launch stream; the dominant kernel alone is in profiles/r02_bench_*_kernel_stats.csv"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, lr, x0, wts)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
