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
