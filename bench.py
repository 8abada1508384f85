#!/usr/bin/env python3
"""bench.py -- MAP gradient iterations/s of the MI355X-native path.

One "step" = one MAP gradient iteration = one ObjectiveFunction::ComputeAllTerms
(data term over all K frames + IRLS-weighted regulariser, cost and gradient,
reference src/optimization/objective_function.cpp:5-20) on device-resident
synthetic inputs.

Workload (--config): cfg2 = BASELINE.json configs[1], the configuration the metric
is quoted on -- 16-frame grayscale, 4x upscale to 2048x2048 HR, Gaussian blur
(3, sigma 1) + BTV (range 3, decay 0.5, lambda 0.01); cfg3 = configs[2] -- 16-frame
RGB, 4x upscale to 4096x4096, BTV.  The SAME workload at every N ("strong" scaling).

N > 1 -- one process per GPU.  `python bench.py --gpus N` without a launcher re-executes
itself under torch.distributed.run.  The path shards inside the C ABI
(srmap_eval_sharded_device: the exchanges are issued by the library on the evaluation's
stream through ITS communicator -- the one RCCL instance of the process; the harness's
own barriers and host scalars travel over a gloo group):
  * rows (default, the scaling path): rank r owns a band of HR rows; before every
    evaluation the boundary rows of x travel to the two neighbours (ncclSend/ncclRecv,
    both directions in one group; --overlap posts them under the interior tile rows);
  * frames (the north-star's exchange, timed in the same run and reported under
    "frames_variant"): rank r owns frames k = r (mod N) and a replica of x; the
    HR gradient and the cost are all-reduced (ncclAllReduce, one group) after the
    local evaluation (reference: alglib_objective.cpp:142-152 is where the gradient
    of all frames meets);
  * --shard channels: N independent channels (the reference's split_channels
    semantics), no collective -- an explicit option only, weak scaling.
A cfg2 run also carries a "cfg3" block: rows strong scaling of configs[2] (16-frame RGB,
4096x4096) with its N = 1 time measured in the same run (by rank 0 alone when N > 1), so
that the N = 1 / 2 / 4 / 8 curve contains a configuration large enough to scale.

Timing: W untimed warm-up steps, then K timed steps between barriers, max over
ranks.  A timed region shorter than --min-timed-ms (50 ms) is extended to whole
multiples of K steps (config.timed_steps); before the warm-up the GPU is driven
for --clock-ramp-ms of untimed evaluations (an MI355X reaches its sustained
clocks only after ~50 ms of load).

Prints ONE JSON line on rank 0 (see the task contract), including
  "roofline":     algorithmic bytes of one step / mean device time per step vs 8 TB/s
                  (+ "hbm_fed": the same with three problems in rotation, i.e. inputs from HBM)
  "cpu_baseline": the CPU oracle (a port of the reference, oracle/) timed on this
                  host on a bounded sample of the same workload: 1 core (the
                  reference is single-threaded) and all cores.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd", "python"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CONFIGS = {
    # name: HR size, channels, frames, scale, blur (ksize, sigma; 0 = none), BTV (range, decay), lambda
    "cfg2": dict(hr=2048, C=1, K=16, s=4, blur=(3, 1.0), btv=(3, 0.5), lam=0.01,
                 label="configs[1]: 16-frame grayscale, 4x upscale to %dx%d, Gaussian blur 3/1.0 + BTV(3,0.5) lambda 0.01, IRLS weights from x0"),
    "cfg3": dict(hr=4096, C=3, K=16, s=4, blur=(0, 0.0), btv=(3, 0.5), lam=0.01,
                 label="configs[2]: 16-frame RGB, 4x upscale to %dx%d, BTV(3,0.5) lambda 0.01, IRLS weights from x0"),
}


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
    """HBM-side bytes per step from the committed PMC profile of this same command (bench.py cannot run rocprofv3
    on itself): the newest profiles/rNN_bench_hbm_pmc.json."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_hbm_pmc.json")), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)[dtype]
            return (2.0 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024.0, os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def limiter_evidence(dtype, dev_ms_per_step):
    """What bounds the dominant kernel according to the newest committed counter profile of this command
    (profiles/rNN_evalz_<dtype>_pmc.json, rocprofv3 --pmc): VALU-busy fraction of the step and the share of their life
    the waves spend parked -- stored counters (bench.py cannot profile itself) against THIS run's step time."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_evalz_%s_pmc.json" % dtype)), reverse=True):
        try:
            with open(path) as f:
                c = json.load(f)["counters_mean_per_launch"]
            # one VALU instruction occupies its SIMD ~1.8 ns at sustained clocks (f64 and select / compare class,
            # profiles/r02_valu_rate.txt); 1024 SIMDs
            per_inst_ns = 1.8 if dtype == "f64" else 1.3   # f32: fma / add ~1.0 ns, select / compare / med3 ~1.8 ns
            valu_us = c["SQ_INSTS_VALU"] / 1024.0 * per_inst_ns * 1e-3
            return {"valu_busy_frac": valu_us / (dev_ms_per_step * 1e3),
                    "wave_parked_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                    "valu_instructions_per_launch": c["SQ_INSTS_VALU"],
                    "note": "valu_busy_frac = SQ_INSTS_VALU / 1024 SIMDs x 1.8 ns (f64) / 1.3 ns (f32) per instruction over this run's device time per "
                            "step; wave_parked_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (barriers and s_waitcnt)",
                    "source": os.path.relpath(path, ROOT)}
        except Exception:
            continue
    return None


def cpu_baseline(cfg, lr, x0, wts, budget_s=7.0):
    """Oracle (CPU restatement of the reference) on a bounded crop of the same workload, scaled by pixel count:
    1 thread (the reference is single-threaded), then one independent crop per core (the image split in tiles)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    s, K = cfg["s"], cfg["K"]
    W = H = cfg["hr"]
    shifts = [[k % s, (k // s) % s] for k in range(K)]

    def make(crop_lr, c0, r0):
        ch = crop_lr * s
        model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=cfg["blur"][0], blur_sigma=cfg["blur"][1])
        prob = orc.Problem(model, lr[:, :1, r0:r0 + crop_lr, c0:c0 + crop_lr])
        prob.add_regularizer(orc.REG_BTV, cfg["lam"], cfg["btv"][0], cfg["btv"][1])
        prob.set_irls_weights(0, wts[:1, r0 * s:r0 * s + ch, c0 * s:c0 * s + ch])
        return prob, np.ascontiguousarray(x0[:1, r0 * s:r0 * s + ch, c0 * s:c0 * s + ch])

    # ---- one core ----
    # single-channel workloads (cfg2): the FULL size, >= 5 evaluations (~2.6 s each); the larger
    # multi-channel configurations keep a crop scaled by pixel count
    full = cfg["C"] == 1 and lr.shape[-1] * s <= 2048
    crop_lr = lr.shape[-1] if full else min(256, lr.shape[-1])  # crop: LR 256x256 -> HR 1024x1024 (~0.65 s per evaluation)
    prob, x = make(crop_lr, 0, 0)
    t0 = time.perf_counter()
    n = 0
    while True:
        prob.objective(x)
        n += 1
        el = time.perf_counter() - t0
        if (el > budget_s and (not full or n >= 5)) or n >= 40:
            break
    per_eval_crop = el / n
    ch = crop_lr * s
    frac = (ch * ch) / float(W * H * cfg["C"])
    per_eval_full = per_eval_crop / frac
    out = {"value": 1.0 / per_eval_full, "unit": "MAP gradient iterations/s", "cores": 1, "kind": "port",
           "sample": ("%d evaluations of the FULL %dx%d workload (%d frames, same blur/BTV), %.2f s each" % (n, ch, ch, K, per_eval_crop))
                     if full else
                     ("%d evaluations of a %dx%d HR crop (%d frames, same blur/BTV), %.2f s each, scaled by pixel count "
                      "x%.0f to the full workload" % (n, ch, ch, K, per_eval_crop, 1 / frac)),
           "ms_per_step": per_eval_full * 1e3}
    # ---- all cores: one tile of the image per thread (ctypes releases the GIL inside the C oracle) ----
    cores = os.cpu_count() or 1
    tile_lr = min(128, lr.shape[-1])
    per_row = max(1, lr.shape[-1] // tile_lr)
    jobs = [make(tile_lr, (i % per_row) * tile_lr, ((i // per_row) % per_row) * tile_lr) for i in range(cores)]
    counts = [0] * cores
    stop = time.perf_counter() + budget_s

    def work(i):
        pr, xx = jobs[i]
        while time.perf_counter() < stop:
            pr.objective(xx)
            counts[i] += 1

    t1 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    el = time.perf_counter() - t1
    tiles_per_s = sum(counts) / el
    tch = tile_lr * s
    full_per_s = tiles_per_s * (tch * tch) / float(W * H * cfg["C"])
    out["all_cores"] = {"value": full_per_s, "unit": "MAP gradient iterations/s", "cores": cores,
                        "sample": "%d threads, each evaluating its own %dx%d HR tile of the image for %.1f s (%d tile "
                                  "evaluations in all), scaled by pixel count to the full workload" % (cores, tch, tch, el, sum(counts)),
                        "ms_per_step": 1e3 / full_per_s if full_per_s > 0 else None}
    # the same figure inside the 1-core record's text (a consumer that keeps only the contract's fields still sees it)
    out["sample"] += "; ALL CORES: %.3f iterations/s on %d hardware threads (one %dx%d HR tile per thread for %.1f s)" % (
        full_per_s, cores, tch, tch, el)
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2")
    ap.add_argument("--min-timed-ms", type=float, default=50.0,
                    help="a timed region shorter than this is extended to whole multiples of --steps")
    ap.add_argument("--clock-ramp-ms", type=float, default=100.0,
                    help="untimed evaluations for this many milliseconds BEFORE the W warm-up steps: the GPU reaches its "
                         "sustained clocks only after ~50 ms of load; 0 disables")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64",
                    help="arithmetic/storage type on device (the reference is f64; f32 is not the parity mode)")
    ap.add_argument("--shard", choices=["rows", "frames", "channels"], default="rows",
                    help="N > 1: rows = HR row bands + halo exchange of x (default, strong scaling; the frame variant is timed "
                         "in the same run); frames = only the frame variant; channels = N independent channels, no "
                         "collective (weak scaling, the reference's split_channels semantics)")
    ap.add_argument("--impl", choices=["auto", "direct", "tiled"], default="auto",
                    help="auto = the library's choice (the 8-row tile kernel where it covers the problem)")
    ap.add_argument("--hr", type=int, default=0, help="override the HR size of the configuration (testing)")
    ap.add_argument("--terms", choices=["all", "data", "reg"], default="all",
                    help="ablation only: evaluate a subset of the objective terms")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-precision", action="store_true", help="skip the labelled block of the other arithmetic type (profiling runs)")
    ap.add_argument("--no-hbm-fed", action="store_true", help="skip the HBM-fed leg of the roofline (N = 1)")
    ap.add_argument("--no-cfg3", action="store_true",
                    help="skip the configs[2] block (rows strong scaling of the 16-frame RGB 4096x4096 problem) that a "
                         "cfg2 run carries so that the N = 1 / 2 / 4 / 8 curve contains a configuration large enough to scale")
    ap.add_argument("--overlap", action="store_true",
                    help="rows, RCCL: post the halo exchange under the interior tile rows (srmap_comm_set_overlap; default: "
                         "exchange first -- the overlapped form is opt-in until it has run on two GPUs)")
    ap.add_argument("--test-single-device", action="store_true",
                    help="testing aid for 1-GPU boxes: all ranks share GPU 0 and talk over gloo through the library's "
                         "host-callback communicator (exercises the N > 1 code paths; the numbers mean nothing)")
    ap.add_argument("--watchdog-s", type=int, default=900, help="N > 1: a rank that has not finished after this many seconds dumps its stack and exits")
    ap.add_argument("--test-rccl-loopback", action="store_true",
                    help="testing aid for 1-GPU boxes: all ranks share GPU 0 and talk over the REAL RCCL communicator -- "
                         "every rank names itself a host of its own (NCCL_HOSTID), so RCCL accepts the shared device and "
                         "runs its socket transport over the loopback interface (the N > 1 RCCL code paths execute; "
                         "the numbers mean nothing)")
    args = ap.parse_args()

    # ---- N > 1 without a launcher: become one (one process per GPU, rendezvous on 127.0.0.1) ----
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner and gloo its connection notes on fd 1
    # when a communicator comes up, so everything but the result goes to stderr from here on
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import srmap
    import srmap_dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))

    import faulthandler

    def watchdog(seconds, what):
        """A rank stuck in an exchange must end the job with an error instead of holding the node.  faulthandler's timer
        is a thread of its own: it fires while the main thread sits inside a native call (a stream synchronisation, a
        collective), where a Python signal handler -- the SIGALRM of rounds 3-4 -- is never run (seen in round 5: a rank
        blocked in hipStreamSynchronize outlived a 900 s alarm)."""
        faulthandler.cancel_dump_traceback_later()
        faulthandler.dump_traceback_later(seconds, exit=True)  # the dumped stack says where (`what` is for the reader)

    if world > 1:
        faulthandler.enable()
        watchdog(args.watchdog_s, "the job")
    one_device = args.test_single_device or args.test_rccl_loopback
    local_rank = 0 if one_device else int(os.environ.get("LOCAL_RANK", "0"))
    if args.test_rccl_loopback and world > 1:  # before anything loads librccl
        os.environ["NCCL_HOSTID"] = "srmap-bench-host-%d" % rank
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
    dist = None
    if world > 1:
        # ONE RCCL instance in the process: the library's own communicator carries every device-side exchange; the
        # harness (barriers, the broadcast of the RCCL unique id, max over ranks of the wall clock) talks gloo on the
        # host.  Round 3 also opened an `nccl` process group here, i.e. a second communicator on PyTorch's RCCL.
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    dtype = srmap.F64 if args.dtype == "f64" else srmap.F32
    tdtype = torch.float64 if args.dtype == "f64" else torch.float32
    E = 8 if args.dtype == "f64" else 4
    terms = {"all": srmap.TERM_ALL, "data": srmap.TERM_DATA, "reg": srmap.TERM_REG}[args.terms]
    impl = {"auto": srmap.IMPL_AUTO, "direct": srmap.IMPL_DIRECT, "tiled": srmap.IMPL_TILED}[args.impl]
    ctx = srmap.Context(local_rank)
    stream = torch.cuda.Stream(device=dev)
    sh = stream.cuda_stream

    # the library's communicator (N > 1): RCCL, or the host callbacks when every rank shares GPU 0
    comm = None
    comm_info = None
    comm_lib = None
    if world > 1 and args.shard != "channels":
        if args.test_single_device:
            comm = srmap.Comm(ctx, rank, world, backend="host", dist=dist)
        else:
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                uid = torch.frombuffer(bytearray(srmap.Comm.unique_id(ctx)), dtype=torch.uint8).clone()
            dist.broadcast(uid, 0)  # gloo, host memory
            watchdog(120, "ncclCommInitRank")
            comm = srmap.Comm(ctx, rank, world, backend="rccl", unique_id=bytes(uid.numpy().tobytes()))
            sys.stderr.write("bench.py: rank %d RCCL communicator up\n" % rank)
            watchdog(args.watchdog_s, "the job")
            if args.overlap:
                comm.set_overlap(True)
        comm_info = comm.info()
        comm_lib = comm.describe()
    is_rccl = comm_info is not None and comm_info[2] == 1

    def barrier():
        stream.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def make_cfg(name):
        cfg = dict(CONFIGS[name])
        if args.hr:
            cfg["hr"] = args.hr
        return cfg

    def run(cfg, shard, steps, warmup, only_rank0=False, n_problems=1, dt=None):
        """Build this rank's part of the workload `cfg` under `shard` and time it.  only_rank0: rank 0 alone evaluates
        the whole problem (the N = 1 reference of a strong-scaling block) while the other ranks wait.  n_problems > 1:
        that many independent copies of the problem evaluated in rotation (the HBM-fed leg: from three copies on a
        problem's 134 MB have left the 256 MiB Infinity Cache when its turn comes again).
        Returns wall seconds per step (max over the ranks that ran), device ms per step on this rank, timed steps, ..."""
        s, K, C = cfg["s"], cfg["K"], cfg["C"]
        W = H = cfg["hr"]
        r_dtype, r_tdtype = (dtype, tdtype) if dt is None else dt   # (the labelled f32 block runs beside an f64 line)
        blur_k, blur_s = cfg["blur"]  # 0 = no blur module
        shifts_all = [[k % s, (k // s) % s] for k in range(K)]
        solo = only_rank0 or world == 1
        if only_rank0 and rank != 0:
            barrier(); barrier(); barrier(); barrier()
            return None
        frame_ids = list(range(K))
        if shard == "frames" and not solo:
            frame_ids = srmap_dist.frame_shard(K, world, rank)
        shifts = [shifts_all[k] for k in frame_ids]
        Hloc, e0, e1, r0, r1 = H, 0, H, 0, H
        bands = None
        if shard == "rows" and not solo:
            halo = srmap_dist.band_halo(s, max(blur_k, 1), s - 1, cfg["btv"][0])
            bands = [srmap_dist.row_band(H, s, world, r, halo) for r in range(world)]
            (r0, r1), (e0, e1) = bands[rank]
            Hloc = e1 - e0
        Cloc = 1 if (shard == "channels" and not solo) else C
        probs = [srmap.Problem(ctx, W, Hloc, Cloc, len(frame_ids), s, shifts, blur_k, blur_s, r_dtype) for _ in range(n_problems)]
        prob = probs[0]
        # synthetic data, seeded; channel = rank under channel sharding
        gt = synth_ground_truth(W, H, max(C, world if shard == "channels" else 1))
        gt = gt[rank:rank + 1] if (shard == "channels" and not solo) else gt[:C]
        gt = gt[:, e0:e1, :]
        lr = np.stack([prob.apply(gt, i) for i in range(len(frame_ids))])
        lr = lr + (5.0 / 255.0) * np.random.default_rng(777 + rank).standard_normal(lr.shape)
        x0 = np.stack([bilinear_upsample(lr[0, c:c + 1], s)[0] for c in range(Cloc)])
        wts = None
        for pr in probs:
            pr.set_impl(impl)
            pr.set_observations(lr)
            reg = pr.add_regularizer(srmap.REG_BTV, cfg["lam"], cfg["btv"][0], cfg["btv"][1])
            if wts is None:
                wts = 1.0 / np.maximum(1e-5, pr.reg_values(reg, x0))
            pr.set_irls_weights(reg, wts)
        sd = None
        if comm is not None and not solo and shard in ("rows", "frames"):
            sd = srmap.ShardDesc()
            if shard == "frames":
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
        xs = [torch.from_numpy(x0).to(dev, r_tdtype).contiguous() for _ in range(n_problems)]
        gs = [torch.empty_like(x) for x in xs]
        turn = [0]

        def step():
            i = turn[0]
            turn[0] = (i + 1) % n_problems
            if sd is not None:
                prob.eval_sharded_device(comm, sd, xs[0].data_ptr(), gs[0].data_ptr(), terms, want_cost=False, stream=sh)
            else:
                probs[i].eval_device(xs[i].data_ptr(), gs[i].data_ptr(), terms, want_cost=False, stream=sh)

        sync_all = (lambda: (stream.synchronize(), torch.cuda.synchronize())) if only_rank0 else barrier
        if sd is not None:  # first contact of this shard mode with the communicator: a step that hangs ends the job in 30 s
            watchdog(30, "the first %s-sharded step" % shard)
            step()
            stream.synchronize()
            watchdog(args.watchdog_s, "the job")
        ramp_steps = 0
        if args.clock_ramp_ms > 0:  # untimed: bring the GPU to its sustained clock state
            # a sharded step is a collective: every rank must issue the SAME number of them, so the decision to go on is
            # taken on the maximum of the ranks' clocks (round 5: each rank consulting its own clock left one rank twenty
            # exchanges ahead of its peers -- a 4-rank RCCL run hung here)
            t_r = time.perf_counter()
            while True:
                el = time.perf_counter() - t_r
                if sd is not None and not only_rank0:
                    el = max_over_ranks(el)
                if el * 1e3 >= args.clock_ramp_ms:
                    break
                for _ in range(20):
                    step()
                stream.synchronize()
                ramp_steps += 20
        sync_all()
        for _ in range(warmup):
            step()
        sync_all()
        # how many steps make --min-timed-ms: from a SYNCHRONISED probe of the steady state (the warm-up's wall time
        # includes two barriers and was, at 20 steps, mostly those)
        probe = max(20, min(200, steps))
        stream.synchronize()
        t_p = time.perf_counter()
        for _ in range(probe):
            step()
        stream.synchronize()
        est = max_over_ranks((time.perf_counter() - t_p) / probe) if not only_rank0 else (time.perf_counter() - t_p) / probe
        reps = 1
        if args.min_timed_ms > 0 and est > 0:
            reps = max(1, int(np.ceil(1.05 * args.min_timed_ms * 1e-3 / (est * steps))))
        n_timed = reps * steps
        sync_all()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            ev0.record(stream)
        for _ in range(n_timed):
            step()
        with torch.cuda.stream(stream):
            ev1.record(stream)
        sync_all()
        wall = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1)  # HIP events on the stream the kernels run on
        if not only_rank0:
            wall = max_over_ranks(wall)
        if only_rank0:  # the waiting ranks sit in four barriers
            barrier(); barrier(); barrier(); barrier()
        return dict(wall_per_step=wall / n_timed, dev_ms_per_step=dev_ms / n_timed, timed_steps=n_timed,
                    ramp_steps=ramp_steps, lr=lr, x0=x0, wts=wts)

    def b_alg_of(cfg, c_unit):
        N, n = cfg["hr"] * cfg["hr"], (cfg["hr"] // cfg["s"]) ** 2
        return E * c_unit * ((2 + 1) * N + cfg["K"] * n)  # x, y, w read once, g written once (per unit)

    cfg = make_cfg(args.config)
    if world == 1:
        main_shard, res = "none", run(cfg, "none", args.steps, args.warmup)
        second = None
    elif args.shard == "channels":
        main_shard, res, second = "channels", run(cfg, "channels", args.steps, args.warmup), None
    elif args.shard == "frames":
        main_shard, res, second = "frames", run(cfg, "frames", args.steps, args.warmup), None
    else:
        main_shard, res = "rows", run(cfg, "rows", args.steps, args.warmup)
        second = run(cfg, "frames", args.steps, args.warmup)

    # ---- HBM-fed leg (N = 1): the same evaluation over three independent problems in rotation ----
    hbm_fed = None
    if world == 1 and not args.no_hbm_fed:
        fed = run(cfg, "none", max(200, args.steps // 4), min(args.warmup, 60), n_problems=3)
        fed_bytes = b_alg_of(cfg, cfg["C"])
        hbm_fed = {"problems_in_rotation": 3, "working_set_bytes": 3 * fed_bytes,
                   "device_ms_per_step": fed["dev_ms_per_step"], "timed_steps": fed["timed_steps"],
                   "achieved": fed_bytes / (fed["dev_ms_per_step"] * 1e-3) / 1e9,
                   "frac": fed_bytes / (fed["dev_ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "note": "three independent copies of the problem evaluated round robin: a copy's inputs have left the 256 MiB "
                           "Infinity Cache when its turn comes again, so every step streams its algorithmic bytes from HBM; "
                           "untimed for `value`"}

    # ---- the other precision on the same clock (N = 1): a labelled extra block, never `value` ----
    other_prec = None
    if world == 1 and args.terms == "all" and not args.no_other_precision:
        o_name = "f32" if args.dtype == "f64" else "f64"
        o_dt = (srmap.F32, torch.float32) if o_name == "f32" else (srmap.F64, torch.float64)
        o_E = 4 if o_name == "f32" else 8
        o = run(cfg, "none", max(200, args.steps // 4), min(args.warmup, 60), dt=o_dt)
        o_bytes = b_alg_of(cfg, cfg["C"]) * o_E / E
        other_prec = {"dtype": o_name, "value": 1.0 / o["wall_per_step"], "unit": "MAP gradient iterations/s",
                      "ms_per_step": o["wall_per_step"] * 1e3, "device_ms_per_step": o["dev_ms_per_step"],
                      "timed_steps": o["timed_steps"], "algorithmic_bytes_per_step": o_bytes,
                      "roofline_frac": o_bytes / (o["dev_ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "note": "same workload, same run, the other arithmetic type (f64 is the parity mode = the reference's "
                              "arithmetic; f32 storage and arithmetic with f64 cost reductions); not part of `value`"}

    # ---- configs[2] block: the configuration of the curve that is large enough to scale (rows, strong scaling) ----
    cfg3_block = None
    if args.config == "cfg2" and not args.no_cfg3 and args.shard != "channels" and args.terms == "all":
        c3 = make_cfg("cfg3")
        st3, wu3 = max(20, args.steps // 10), max(5, args.warmup // 10)
        ref3 = run(c3, "none", st3, wu3, only_rank0=(world > 1))       # the N = 1 time, measured in this very run
        sh3 = run(c3, "rows", st3, wu3) if world > 1 else ref3
        if rank == 0:
            cfg3_block = {"workload": c3["label"] % (c3["hr"], c3["hr"]), "shard": "rows" if world > 1 else "none",
                          "scaling": "strong", "value": 1.0 / sh3["wall_per_step"], "unit": "MAP gradient iterations/s",
                          "ms_per_step": sh3["wall_per_step"] * 1e3, "device_ms_per_step": sh3["dev_ms_per_step"],
                          "timed_steps": sh3["timed_steps"],
                          "n1_reference_ms_per_step": ref3["wall_per_step"] * 1e3,
                          "speedup_vs_n1_in_this_run": ref3["wall_per_step"] / sh3["wall_per_step"],
                          "roofline_frac_n1": b_alg_of(c3, c3["C"]) / (ref3["dev_ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS}

    if rank == 0:
        s, K, C = cfg["s"], cfg["K"], cfg["C"]
        W = H = cfg["hr"]
        units = float(world) if main_shard == "channels" else 1.0
        c_unit = 1 if main_shard == "channels" else C
        b_alg = b_alg_of(cfg, c_unit)
        step_dev_s = res["dev_ms_per_step"] * 1e-3
        share = 1.0 if main_shard in ("none", "channels") else 1.0 / world  # bytes this GPU moves per step
        achieved = b_alg * share / step_dev_s / 1e9
        traffic, traffic_src = pmc_traffic(args.dtype)
        sr_name, ar_name = ("ncclSend/ncclRecv", "ncclAllReduce") if (is_rccl or comm_info is None) else \
            ("host-callback send/recv (test backend)", "host-callback all-reduce (test backend)")
        collective = {"none": "none",
                      "rows": sr_name + " of the halo rows of x (both directions in one group) inside srmap_eval_sharded_device",
                      "frames": ar_name + "(g, C*N) + " + ar_name + "(cost) in one group inside srmap_eval_sharded_device",
                      "channels": "none (split_channels: independent per-channel solves)"}
        out = {
            "metric": "MAP gradient iterations/sec at fixed HR size",
            "value": units / res["wall_per_step"],
            "unit": "MAP gradient iterations/s" if main_shard != "channels" else "channel-iterations/s (one %d-frame %dx%d channel per GPU)" % (K, W, H),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["wall_per_step"] * 1e3, "higher_is_better": True,
            # ONE label for the whole N = 1 / 2 / 4 / 8 curve: the workload is the same at every N (only --shard channels
            # gives every GPU its own channel)
            "scaling": "weak" if main_shard == "channels" or (world == 1 and args.shard == "channels") else "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": cfg["label"] % (W, H), "frames": K, "scale": s, "channels": C,
                       "shard": main_shard, "collective_per_step": collective[main_shard],
                       "comm_ranks": (comm_info[1] if comm_info else (world if world > 1 else 1)),
                       "comm_backend": (None if comm_info is None else
                                        (("rccl over loopback sockets, all ranks on GPU 0 (test)" if args.test_rccl_loopback else "rccl")
                                         if is_rccl else "host callbacks (test)")),
                       "comm_library": comm_lib, "harness_group": ("gloo" if world > 1 else None),
                       "halo_overlap": bool(args.overlap) if main_shard == "rows" else None,
                       "impl": args.impl, "device_ms_per_step": res["dev_ms_per_step"],
                       "timed_steps": res["timed_steps"], "clock_ramp_steps_before_warmup": res["ramp_steps"],
                       "parity_mode": args.dtype == "f64"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # `bound` names the roofline the path is PRICED against (its algorithmic bytes over HBM); what
                         # limits the kernel at this size is not bandwidth: counter traffic ~1.05x algorithmic at < half of
                         # peak -- the waves are parked (barriers, request and LDS waits) a third of their life
                         "limited_by": "latency/occupancy (not bandwidth)",
                         "limiter_evidence": limiter_evidence(args.dtype, res["dev_ms_per_step"]),
                         "residency": "one problem re-evaluated every step: its %.0f MB working set stays in the 256 MiB Infinity "
                                      "Cache between steps (the contract's step); `hbm_fed` is the same step with the inputs "
                                      "coming from HBM" % (b_alg * share / 1e6),
                         "hbm_fed": hbm_fed,
                         "traffic_source": (traffic_src + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of this "
                                            "command; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, the factor 2 being the gfx950 "
                                            "FETCH_SIZE correction of MI355X_MICROARCH.md") if traffic_src else None,
                         "algorithmic_bytes_per_step": b_alg * share,
                         "kernel": "whole evaluation (every kernel of one step: k_eval_z with its in-kernel reduction), HIP "
                                   "events on the launch stream; the dominant kernel alone is in profiles/r06_bench_*_kernel_stats.csv"},
        }
        if second is not None:
            out["frames_variant"] = {"value": 1.0 / second["wall_per_step"], "unit": "MAP gradient iterations/s",
                                     "ms_per_step": second["wall_per_step"] * 1e3, "scaling": "strong",
                                     "device_ms_per_step": second["dev_ms_per_step"], "timed_steps": second["timed_steps"],
                                     "collective_per_step": collective["frames"]}
        if other_prec is not None:
            out[other_prec["dtype"]] = other_prec
        if cfg3_block is not None:
            out["cfg3"] = cfg3_block
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, res["lr"], res["x0"], res["wts"])
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
