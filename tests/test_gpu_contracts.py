"""Contracts of the C ABI stated in include/srmap.h: the input domain of the tile kernels (the x * 2^Q staging), the
stream-ordering rules of the device-pointer entry points, and the report of the kernel family in use."""
import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sr():
    import srmap
    return srmap


@pytest.fixture(scope="module")
def ctx(sr):
    return sr.Context(0)


from parity_log import relerr  # noqa: E402  (max |a - ref| / max(1, |ref|), logged when SRMAP_PARITY_LOG is set)


@pytest.mark.parametrize("log2_scale", [21, 300, -300, 0])
@pytest.mark.parametrize("reg", ["btv", "tv"])
def test_tile_path_input_domain_f64(sr, ctx, log2_scale, reg):
    """Pixel values of magnitude 2^21 (beyond the old 2^20 limit of the x * 2^1000 staging), 2^300 and 2^-300: the
    tile kernels (x staged times 2^512) agree with the direct kernels and with the oracle.  The objective is
    homogeneous -- data term degree 2, regulariser degree 1 in (x, y) -- so the comparison is relative."""
    rng = np.random.default_rng(5)
    s, K, h, w = 4, 16, 24, 40
    H, W = h * s, w * s
    scale = float(2.0 ** log2_scale)
    shifts = [[k % s, (k // s) % s] for k in range(K)]
    lr = rng.random((K, 1, h, w)) * scale
    x = (np.round(rng.random((1, H, W)) * 64) / 64) * scale   # plenty of exactly equal neighbours (sign(0) taps)
    wts = 0.5 + rng.random((1, H, W))
    model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=3, blur_sigma=1.0)
    ref = orc.Problem(model, lr)
    args = (orc.REG_BTV, 0.01 * scale, 3, 0.5) if reg == "btv" else (orc.REG_TV, 0.01 * scale, 0, 0.0)
    ref.add_regularizer(*args)
    ref.set_irls_weights(0, wts)
    f_ref, g_ref = ref.objective(x)
    p = sr.Problem(ctx, W, H, 1, K, s, shifts, 3, 1.0, sr.F64)
    p.set_observations(lr)
    p.set_irls_weights(p.add_regularizer(*args), wts)
    for impl in (sr.IMPL_TILED, sr.IMPL_DIRECT):
        p.set_impl(impl)
        assert p.active_impl() == impl
        f, g = p.eval(x)
        assert np.isfinite(f) and np.all(np.isfinite(g))
        assert abs(f - f_ref) <= 1e-12 * abs(f_ref)
        assert np.max(np.abs(g - g_ref)) <= 1e-12 * np.max(np.abs(g_ref))


def test_tile_path_input_domain_f32(sr, ctx):
    """f32 storage: magnitudes 2^21 and 2^40 (the old limit was 2^24), tile against direct kernels."""
    rng = np.random.default_rng(6)
    s, K, h, w = 2, 4, 32, 48
    H, W = h * s, w * s
    shifts = [[0, 0], [1, 1], [0, 1], [1, 0]]
    for log2_scale in (21, 40):
        scale = float(2.0 ** log2_scale)
        lr = rng.random((K, 1, h, w)) * scale
        x = (np.round(rng.random((1, H, W)) * 64) / 64) * scale
        p = sr.Problem(ctx, W, H, 1, K, s, shifts, 3, 1.0, sr.F32)
        p.set_observations(lr)
        p.add_regularizer(sr.REG_BTV, 0.01 * scale, 2, 0.7)
        res = {}
        for impl in (sr.IMPL_TILED, sr.IMPL_DIRECT):
            p.set_impl(impl)
            res[impl] = p.eval(x)
        (f1, g1), (f2, g2) = res[sr.IMPL_TILED], res[sr.IMPL_DIRECT]
        assert np.isfinite(f1) and abs(f1 - f2) <= 1e-5 * abs(f2)
        assert np.max(np.abs(g1 - g2)) <= 2e-5 * np.max(np.abs(g2))


def test_weights_written_on_one_stream_are_seen_by_an_evaluation_on_another(sr, ctx):
    """srmap_update_irls_weights_device on stream A (not waited for), srmap_eval_device on stream B: the library orders
    B after the write (an event), and a later re-write on A after the evaluations on B."""
    import torch
    W = H = 1024
    s, K = 4, 16
    shifts = [[k % s, (k // s) % s] for k in range(K)]
    p = sr.Problem(ctx, W, H, 1, K, s, shifts, 3, 1.0, sr.F64)
    g0 = torch.Generator(device="cuda").manual_seed(1)
    y = torch.rand((K, 1, H // s, W // s), dtype=torch.float64, device="cuda", generator=g0)
    xa = torch.rand((1, H, W), dtype=torch.float64, device="cuda", generator=g0)
    xb = torch.rand((1, H, W), dtype=torch.float64, device="cuda", generator=g0)
    torch.cuda.synchronize()
    A, B = torch.cuda.Stream(), torch.cuda.Stream()
    p.set_observations_device(y.data_ptr(), stream=A.cuda_stream)
    r = p.add_regularizer(sr.REG_BTV, 0.01, 3, 0.5)
    ga, gb = torch.empty_like(xa), torch.empty_like(xa)

    def sequential(xw):
        p.update_irls_weights_device(r, xw.data_ptr())
        ctx.synchronize()
        g = torch.empty_like(xa)
        f = p.eval_device(xa.data_ptr(), g.data_ptr(), sr.TERM_ALL, want_cost=True)
        return f, g.clone()

    fa_ref, ga_ref = sequential(xa)
    fb_ref, gb_ref = sequential(xb)
    assert not torch.equal(ga_ref, gb_ref)
    for _ in range(20):
        # a long queue on A in front of the weight update, so an unordered B would run first
        for _k in range(10):
            p.update_irls_weights_device(r, xb.data_ptr(), stream=A.cuda_stream)
        p.update_irls_weights_device(r, xa.data_ptr(), stream=A.cuda_stream)
        fa = p.eval_device(xa.data_ptr(), ga.data_ptr(), sr.TERM_ALL, want_cost=True, stream=B.cuda_stream)
        # 30 evaluations in flight on B while A re-writes the weights: the writer drains B first
        for _k in range(30):
            p.eval_device(xa.data_ptr(), ga.data_ptr(), sr.TERM_ALL, want_cost=False, stream=B.cuda_stream)
        p.update_irls_weights_device(r, xb.data_ptr(), stream=A.cuda_stream)
        fb = p.eval_device(xa.data_ptr(), gb.data_ptr(), sr.TERM_ALL, want_cost=True, stream=B.cuda_stream)
        torch.cuda.synchronize()
        assert fa == fa_ref and torch.equal(ga, ga_ref)
        assert fb == fb_ref and torch.equal(gb, gb_ref)


def test_pca_on_the_callers_stream(sr, ctx):
    """srmap_channel_pca_device takes the stream its input was produced on."""
    import torch
    S = torch.cuda.Stream()
    rows, n = 12, 200000
    with torch.cuda.stream(S):
        base = torch.rand((rows, n), dtype=torch.float64, device="cuda")
        for _ in range(20):
            base = base * 0.999 + 0.001 * torch.roll(base, 1, 0)   # a queue of producers on S
        mean, ev, basis = ctx.pca_device(base.data_ptr(), rows, n, 0, 1, n, stream=S.cuda_stream)
    host = base.cpu().numpy()
    assert np.allclose(mean, host.mean(axis=1), atol=1e-12)
    cov = np.cov(host, bias=True)
    assert np.allclose(ev, np.sort(np.linalg.eigvalsh(cov))[::-1], atol=1e-12)
    assert np.allclose(basis @ cov @ basis.T, np.diag(ev), atol=1e-12)


def test_active_impl_reports_the_fallback(sr, ctx):
    shifts = [[0, 0], [1, 1], [0, 1], [1, 0]]
    p = sr.Problem(ctx, 64, 48, 1, 4, 2, shifts, 3, 1.0, sr.F64)
    assert p.active_impl() == sr.IMPL_TILED
    p.set_impl(sr.IMPL_DIRECT)
    assert p.active_impl() == sr.IMPL_DIRECT
    q = sr.Problem(ctx, 64, 48, 1, 4, 2, shifts, 5, 1.0, sr.F64)  # blur size 5: outside the tile kernels' coverage
    assert q.active_impl() == sr.IMPL_DIRECT
