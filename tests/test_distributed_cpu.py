"""world_size-2 gloo tests of the N > 1 path (runs on CPU): the channel- and
frame-sharded objective assembled by srmap_dist.ShardedObjective equals the
single-process objective.  The local evaluator here is the CPU oracle; on the
GPU box the same class wraps srmap.Problem.eval_device over RCCL (bench.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "super-resolution_amd", "python")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _problem_inputs():
    rng = np.random.default_rng(123)
    s, K, C, h, w = 2, 6, 4, 6, 5
    shifts = [[0, 0], [1, 1], [0, 1], [1, 0], [-1, 0], [2, -1]]
    lr = rng.random((K, C, h, w))
    x = rng.random((C, h * s, w * s))
    wts = 0.5 + rng.random((C, h * s, w * s))
    return s, K, C, h, w, shifts, lr, x, wts


def _worker(rank, world, port, mode, out_q):
    _setup_paths()
    import oracle as orc
    import srmap_dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, K, C, h, w, shifts, lr, x, wts = _problem_inputs()
    if mode == "frames":
        ks = srmap_dist.frame_shard(K, world, rank)
        model = orc.ImageModel(scale=s, shifts=[shifts[k] for k in ks], blur_ksize=3, blur_sigma=1.0)
        prob = orc.Problem(model, lr[ks])
        prob.add_regularizer(orc.REG_BTV, 0.02, 2, 0.5)
        prob.set_irls_weights(0, wts)

        def local_eval(xv, terms):
            f, g = (0.0, np.zeros(xv.size))
            if terms & 1:
                fd, gd = prob.data_term(xv)
                f, g = f + fd, g + np.ravel(gd)
            if terms & 2:
                fr, gr = prob.reg_term(0, xv)
                f, g = f + fr, g + np.ravel(gr)
            return f, torch.from_numpy(g)
        obj = srmap_dist.ShardedObjective("frames", local_eval, dist)
        cost, grad = obj.eval(x)
        out_q.put((rank, cost, grad.numpy().copy(), None))
    else:
        c0, c1 = srmap_dist.channel_shard(C, world, rank)
        model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=3, blur_sigma=1.0)
        prob = orc.Problem(model, lr[:, c0:c1])
        prob.add_regularizer(orc.REG_BTV, 0.02, 2, 0.5)
        prob.set_irls_weights(0, wts[c0:c1])

        def local_eval(xv, terms):
            f, g = prob.objective(xv)
            return f, torch.from_numpy(np.ravel(g).copy())
        obj = srmap_dist.ShardedObjective("channels", local_eval, dist)
        cost, grad = obj.eval(x[c0:c1])
        out_q.put((rank, cost, grad.numpy().copy(), (c0, c1)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["channels", "frames"])
def test_sharded_objective_world2(mode):
    _setup_paths()
    import oracle as orc
    s, K, C, h, w, shifts, lr, x, wts = _problem_inputs()
    model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=3, blur_sigma=1.0)
    ref = orc.Problem(model, lr)
    ref.add_regularizer(orc.REG_BTV, 0.02, 2, 0.5)
    ref.set_irls_weights(0, wts)
    f_ref, g_ref = ref.objective(x)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + (0 if mode == "channels" else 600)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=60) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g_ref = g_ref.reshape(C, -1)
    for rank, cost, grad, rng_ in results:
        assert cost == pytest.approx(f_ref, rel=1e-12)
        if mode == "frames":
            assert np.allclose(grad.reshape(C, -1), g_ref, rtol=1e-12, atol=1e-13)
        else:
            c0, c1 = rng_
            assert np.allclose(grad.reshape(c1 - c0, -1), g_ref[c0:c1], rtol=1e-12, atol=1e-13)


def test_shard_helpers_cover_everything():
    _setup_paths()
    import srmap_dist
    for world in (1, 2, 3, 4, 8):
        for n in (1, 3, 8, 16, 17, 128):
            frames = sorted(k for r in range(world) for k in srmap_dist.frame_shard(n, world, r))
            assert frames == list(range(n))
            blocks = [srmap_dist.channel_shard(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))


# ---------------------------------------------------------------------------
# row bands (spatial sharding) and the 2-D frames x channels grid
# ---------------------------------------------------------------------------
def _band_inputs():
    rng = np.random.default_rng(321)
    s, K, C, h, w = 2, 4, 2, 24, 7
    shifts = [[0, 0], [1, 1], [-1, 2], [2, -1]]
    lr = rng.random((K, C, h, w))
    x = rng.random((C, h * s, w * s))
    wts = 0.5 + rng.random((C, h * s, w * s))
    return s, K, C, h, w, shifts, lr, x, wts


def _band_worker(rank, world, port, out_q):
    _setup_paths()
    import oracle as orc
    import srmap_dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, K, C, h, w, shifts, lr, x, wts = _band_inputs()
    H = h * s
    lam, R, alpha = 0.02, 2, 0.5
    halo = srmap_dist.band_halo(s, 3, 2, R)
    bands = [srmap_dist.row_band(H, s, world, r, halo) for r in range(world)]
    (r0, r1), (e0, e1) = bands[rank]
    model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=3, blur_sigma=1.0)
    y_band = lr[:, :, e0 // s:e1 // s, :]
    prob = orc.Problem(model, y_band)
    prob.add_regularizer(orc.REG_BTV, lam, R, alpha)
    w_band = wts[:, e0:e1, :]
    prob.set_irls_weights(0, w_band)

    def local_eval(xb):
        xn = xb.numpy()
        _, g = prob.objective(xn)
        # cost of the owned rows only (what srmap_problem_set_cost_rows does on the GPU)
        i0, i1 = (r0 - e0) // s, (r1 - e0) // s
        fd = sum(float(np.sum((model.apply(xn, k) - y_band[k])[:, i0:i1, :] ** 2)) for k in range(K)) * s * s
        rv = orc.reg_values(orc.REG_BTV, xn, R, alpha)
        fr = lam * float(np.sum((w_band * rv * rv)[:, r0 - e0:r1 - e0, :]))
        return fd + fr, torch.from_numpy(np.asarray(g).reshape(xn.shape).copy())

    obj = srmap_dist.BandObjective((r0, r1), (e0, e1), local_eval, dist)
    obj.set_peer_halos([(b[0][0] - b[1][0], b[1][1] - b[0][1]) for b in bands])
    xb = torch.from_numpy(x[:, e0:e1, :].copy())
    # the halo rows start out stale: exchange_halos must refresh them from the neighbours
    xb[:, :r0 - e0, :] = -7.0
    xb[:, r1 - e0:, :] = -7.0
    cost, g_owned = obj.eval(xb)
    out_q.put((rank, cost, g_owned.numpy().copy(), (r0, r1)))
    dist.barrier()
    dist.destroy_process_group()


def test_row_band_objective_world2():
    _setup_paths()
    import oracle as orc
    s, K, C, h, w, shifts, lr, x, wts = _band_inputs()
    model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=3, blur_sigma=1.0)
    ref = orc.Problem(model, lr)
    ref.add_regularizer(orc.REG_BTV, 0.02, 2, 0.5)
    ref.set_irls_weights(0, wts)
    f_ref, g_ref = ref.objective(x)
    g_ref = np.asarray(g_ref).reshape(x.shape)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30700 + (os.getpid() % 500)
    procs = [ctx.Process(target=_band_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=60) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, cost, g_owned, (r0, r1) in results:
        assert cost == pytest.approx(f_ref, rel=1e-12)
        assert np.allclose(g_owned, g_ref[:, r0:r1, :], rtol=1e-12, atol=1e-13)


def _grid_worker(rank, world, port, frame_groups, out_q):
    _setup_paths()
    import oracle as orc
    import srmap_dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, K, C, h, w, shifts, lr, x, wts = _problem_inputs()
    cg, fg = srmap_dist.grid_coords(world, rank, frame_groups)
    channel_groups = world // frame_groups
    # every rank creates every group, in the same order (torch.distributed requirement)
    groups = [dist.new_group(list(range(c * frame_groups, (c + 1) * frame_groups))) for c in range(channel_groups)]
    c0, c1 = srmap_dist.channel_shard(C, channel_groups, cg)
    ks = srmap_dist.frame_shard(K, frame_groups, fg)
    model = orc.ImageModel(scale=s, shifts=[shifts[k] for k in ks], blur_ksize=3, blur_sigma=1.0)
    prob = orc.Problem(model, lr[ks][:, c0:c1])
    prob.add_regularizer(orc.REG_BTV, 0.02, 2, 0.5)
    prob.set_irls_weights(0, wts[c0:c1])

    def local_eval(xv, terms):
        f, g = (0.0, np.zeros(xv.size))
        if terms & 1:
            fd, gd = prob.data_term(xv)
            f, g = f + fd, g + np.ravel(gd)
        if terms & 2:
            fr, gr = prob.reg_term(0, xv)
            f, g = f + fr, g + np.ravel(gr)
        return f, torch.from_numpy(g)
    obj = srmap_dist.ShardedObjective("grid", local_eval, dist, frame_groups=frame_groups, channel_group=groups[cg])
    cost, grad = obj.eval(x[c0:c1])
    out_q.put((rank, cost, grad.numpy().copy(), (c0, c1)))
    dist.barrier()
    dist.destroy_process_group()


def test_grid_objective_world4():
    """frames x channels: 2 channel blocks x 2 frame shards."""
    _setup_paths()
    import oracle as orc
    s, K, C, h, w, shifts, lr, x, wts = _problem_inputs()
    model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=3, blur_sigma=1.0)
    ref = orc.Problem(model, lr)
    ref.add_regularizer(orc.REG_BTV, 0.02, 2, 0.5)
    ref.set_irls_weights(0, wts)
    f_ref, g_ref = ref.objective(x)
    g_ref = g_ref.reshape(C, -1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31900 + (os.getpid() % 500)
    procs = [ctx.Process(target=_grid_worker, args=(r, 4, port, 2, q)) for r in range(4)]
    for p in procs:
        p.start()
    results = [q.get(timeout=90) for _ in range(4)]
    for p in procs:
        p.join(timeout=90)
        assert p.exitcode == 0
    for rank, cost, grad, (c0, c1) in results:
        assert cost == pytest.approx(f_ref, rel=1e-12)
        assert np.allclose(grad.reshape(c1 - c0, -1), g_ref[c0:c1], rtol=1e-12, atol=1e-13)


def test_row_band_helpers():
    _setup_paths()
    import srmap_dist
    for H, s in ((2048, 4), (1023 // 3 * 3, 3), (96, 2)):
        for world in (1, 2, 3, 8):
            halo = srmap_dist.band_halo(s, 3, 3, 3)
            assert halo % s == 0 and halo >= 8
            bands = [srmap_dist.row_band(H, s, world, r, halo) for r in range(world)]
            assert bands[0][0][0] == 0 and bands[-1][0][1] == H
            for r in range(world):
                (r0, r1), (e0, e1) = bands[r]
                assert r0 % s == 0 and e0 % s == 0 and 0 <= e0 <= r0 < r1 <= e1 <= H
                if r + 1 < world:
                    assert r1 == bands[r + 1][0][0]
