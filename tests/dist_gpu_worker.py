"""One rank of the multi-process sharded tests (tests/test_gpu_sharded.py): every rank shares GPU 0, talks over gloo
(through the library's host-callback communicator) and runs its shard of the joint evaluation / solve through the C
ABI; rank 0 also evaluates the un-sharded problem and writes the comparison as JSON.

    python dist_gpu_worker.py <rank> <world> <port> <mode> <out.json> [host|rccl]

"rccl": the same shards over the RCCL communicator.  RCCL refuses two ranks on one device of one host, so every rank
names itself a host of its own (NCCL_HOSTID) and RCCL runs its socket transport over the loopback interface: not the
xGMI path, but ncclCommInitRank with N > 1, ncclAllReduce, the grouped ncclSend / ncclRecv halo exchange and
ncclCommSplit execute for real.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "super-resolution_amd", "python"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


from parity_log import relerr  # noqa: E402  (max |a - ref| / max(1, |ref|), logged when SRMAP_PARITY_LOG is set)


def main():
    rank, world, port, mode, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    backend = sys.argv[6] if len(sys.argv) > 6 else "host"
    if backend == "rccl":  # before anything loads librccl
        os.environ["NCCL_HOSTID"] = "srmap-test-host-%d" % rank
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
    torch.cuda.init()
    torch.zeros(1, device="cuda")  # torch's HIP runtime first (see tests/conftest.py)
    import srmap
    import srmap_dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ctx = srmap.Context(0)
    if backend == "rccl":
        box = [srmap.Comm.unique_id(ctx) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = srmap.Comm(ctx, rank, world, backend="rccl", unique_id=box[0])
        print("RCCL_COMM_OK rank %d" % rank, flush=True)  # past this line a failure is the library's, not the environment's
    else:
        comm = srmap.Comm(ctx, rank, world, backend="host", dist=dist)

    rng = np.random.default_rng(123)
    s, b, sigma = 4, 3, 1.0
    K = 8
    shifts = [[k % s, (k * 3) % s] for k in range(K)]
    C = 4 if mode in ("channels", "grid") else 2
    W, H = 80, 96
    w, h = W // s, H // s
    gt = rng.random((C, H, W))
    lr = rng.random((K, C, h, w))
    x0 = rng.random((C, H, W))
    regs = [(srmap.REG_BTV, 0.02, 3, 0.5)] + ([(srmap.REG_TV3D, 0.03, 0, 0.0)] if mode in ("channels", "grid") else [])
    if mode == "frames2":  # a second regulariser: the direct kernels join in, the regulariser stays on reg_rank (no band split)
        regs.append((srmap.REG_TV, 0.01, 0, 0.0))
        mode = "frames"
    if mode == "rows_overlap":  # the halo exchange posted under the interior tile rows (srmap_comm_set_overlap)
        comm.set_overlap(True)
        mode = "rows"
    force_direct = False
    if mode == "frames_mixed":  # ONE rank on the direct kernels (it cannot evaluate a row band): the ranks must agree on reg_rank
        force_direct = rank == 1
        mode = "frames"
    opts = srmap.default_irls_options()
    opts.max_num_irls_iterations = 2
    opts.max_num_solver_iterations = 6

    def full_problem():
        p = srmap.Problem(ctx, W, H, C, K, s, shifts, b, sigma, srmap.F64)
        p.set_observations(lr)
        for r in regs:
            p.add_regularizer(*r)
        return p

    sd = srmap.ShardDesc()
    if mode == "frames":
        ids = srmap_dist.frame_shard(K, world, rank)
        p = srmap.Problem(ctx, W, H, C, len(ids), s, [shifts[k] for k in ids], b, sigma, srmap.F64)
        p.set_observations(lr[ids])
        for r in regs:
            p.add_regularizer(*r)
        if force_direct:
            p.set_impl(srmap.IMPL_DIRECT)
        sd.mode, sd.reg_rank = srmap.SHARD_FRAMES, 0
        x_loc = x0
        own = (slice(None), slice(None))
    elif mode == "rows":
        halo = srmap_dist.band_halo(s, b, s - 1, 3)
        bands = [srmap_dist.row_band(H, s, world, r, halo) for r in range(world)]
        (r0, r1), (e0, e1) = bands[rank]
        p = srmap.Problem(ctx, W, e1 - e0, C, K, s, shifts, b, sigma, srmap.F64)
        p.set_observations(lr[:, :, e0 // s:e1 // s, :])
        for r in regs:
            p.add_regularizer(*r)
        p.set_cost_rows(r0 - e0, r1 - e0)
        sd.mode = srmap.SHARD_ROWS
        sd.own_row0, sd.own_row1 = r0 - e0, r1 - e0
        if rank + 1 < world:
            (n0, n1), (ne0, ne1) = bands[rank + 1]
            sd.send_down_rows = n0 - ne0
        if rank > 0:
            (u0, u1), (ue0, ue1) = bands[rank - 1]
            sd.send_up_rows = ue1 - u1
        x_loc = x0[:, e0:e1, :].copy()
        # stale halos: the library must refresh them before the first evaluation
        if r0 > e0:
            x_loc[:, :r0 - e0, :] = -7.0
        if e1 > r1:
            x_loc[:, r1 - e0:, :] = -7.0
        own = (slice(None), slice(r0, r1))
        loc_rows = (r0 - e0, r1 - e0)
    elif mode == "grid":  # frames x channels (BASELINE configs[4]): rank = channel block * frame_groups + frame group
        FG = 2
        nblocks = world // FG
        cb, fg = rank // FG, rank % FG
        c0, c1 = srmap_dist.channel_shard(C, nblocks, cb)
        lo, hi = (1 if c0 > 0 else 0), (1 if c1 < C else 0)
        ids = srmap_dist.frame_shard(K, FG, fg)
        p = srmap.Problem(ctx, W, H, (c1 - c0) + lo + hi, len(ids), s, [shifts[k] for k in ids], b, sigma, srmap.F64)
        p.set_observations(lr[ids][:, c0 - lo:c1 + hi])
        for r in regs:
            p.add_regularizer(*r)
        if backend == "rccl":
            fcomm = comm.split(cb, fg, fg, FG)  # ncclCommSplit: the frame groups of one channel block
        else:
            groups = [dist.new_group([blk * FG + f for f in range(FG)]) for blk in range(nblocks)]  # collective: every rank creates all
            fcomm = srmap.Comm(ctx, fg, FG, backend="host", dist=dist, group=groups[cb], group_ranks=[cb * FG + f for f in range(FG)])
        sd.mode = srmap.SHARD_GRID
        sd.own_ch0, sd.own_ch1 = lo, lo + (c1 - c0)
        sd.frame_groups = FG
        sd.frame_comm = fcomm._h
        x_loc = x0[c0 - lo:c1 + hi].copy()
        if lo:
            x_loc[0] = -7.0
        if hi:
            x_loc[-1] = -7.0
        own = (slice(c0, c1), slice(None))
    else:  # channels, coupled by the 3-D TV regulariser: one halo plane per neighbour
        c0, c1 = srmap_dist.channel_shard(C, world, rank)
        lo, hi = (1 if c0 > 0 else 0), (1 if c1 < C else 0)
        p = srmap.Problem(ctx, W, H, (c1 - c0) + lo + hi, K, s, shifts, b, sigma, srmap.F64)
        p.set_observations(lr[:, c0 - lo:c1 + hi])
        for r in regs:
            p.add_regularizer(*r)
        sd.mode = srmap.SHARD_CHANNELS
        sd.own_ch0, sd.own_ch1 = lo, lo + (c1 - c0)
        x_loc = x0[c0 - lo:c1 + hi].copy()
        if lo:
            x_loc[0] = -7.0
        if hi:
            x_loc[-1] = -7.0
        own = (slice(c0, c1), slice(None))

    # ---- one sharded evaluation against the un-sharded one (weights = ones) ----
    xd = torch.from_numpy(np.ascontiguousarray(x_loc)).cuda()
    gd = torch.zeros_like(xd)
    f = p.eval_sharded_device(comm, sd, xd.data_ptr(), gd.data_ptr(), srmap.TERM_ALL, want_cost=True)
    torch.cuda.synchronize()
    g_loc = gd.cpu().numpy()
    res = {"mode": mode, "backend": comm.describe()}
    if mode == "frames":
        g_own = g_loc
    elif mode == "rows":
        g_own = g_loc[:, loc_rows[0]:loc_rows[1], :]
    else:
        g_own = g_loc[sd.own_ch0:sd.own_ch1]
    # ---- the sharded solve ----
    x_sol, rep = p.solve(x_loc, opts, comm=comm, shard=sd)
    if mode == "frames":
        x_own = x_sol
    elif mode == "rows":
        x_own = x_sol[:, loc_rows[0]:loc_rows[1], :]
    else:
        x_own = x_sol[sd.own_ch0:sd.own_ch1]
    gathered_g = [None] * world
    gathered_x = [None] * world
    dist.all_gather_object(gathered_g, g_own)
    dist.all_gather_object(gathered_x, x_own)
    if rank == 0:
        full = full_problem()
        f_ref, g_ref = full.eval(x0)
        x_ref, rep_ref = full.solve(x0, opts)
        if mode == "frames":
            g_all, x_all = gathered_g[0], gathered_x[0]
            res["replicas_equal"] = all(np.array_equal(gathered_x[0], gx) for gx in gathered_x)
        elif mode == "rows":
            g_all, x_all = np.concatenate(gathered_g, axis=1), np.concatenate(gathered_x, axis=1)
        elif mode == "grid":  # one replica per channel block (frame group 0); the replicas of a block must agree
            g_all = np.concatenate(gathered_g[0::2], axis=0)
            x_all = np.concatenate(gathered_x[0::2], axis=0)
            res["replicas_equal"] = all(np.array_equal(gathered_x[i], gathered_x[i + 1]) and
                                        np.array_equal(gathered_g[i], gathered_g[i + 1]) for i in range(0, world, 2))
        else:
            g_all, x_all = np.concatenate(gathered_g, axis=0), np.concatenate(gathered_x, axis=0)
        res.update(cost_err=abs(f - f_ref) / max(1.0, abs(f_ref)), grad_err=relerr(g_all, g_ref),
                   solve_err=relerr(x_all, x_ref), evals=[rep.evaluations, rep_ref.evaluations],
                   cg=[rep.cg_iterations, rep_ref.cg_iterations], irls=[rep.irls_rounds, rep_ref.irls_rounds],
                   final_cost=[rep.final_cost, rep_ref.final_cost])
        with open(out, "w") as fo:
            json.dump(res, fo)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
