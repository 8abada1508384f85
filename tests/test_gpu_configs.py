"""The five BASELINE.json configurations through the C ABI on the GPU.

For each of cfg3 / cfg4 / cfg5 (and cfg2):
  (i)  a reduced-size replica with the STATED frame count, scale, regulariser
       mix and >= 3 channels against the CPU oracle (f64 1e-12 per element,
       f32 2e-5), and
  (ii) the full-size configuration on device-generated data: the LDS-tiled path
       against the independent direct kernels, and both against a whole-array
       torch (f64, on the GPU) restatement of the formulas of SURVEY.md section
       8(a') -- shifted slices, no per-pixel loops, written independently of
       both the C oracle and the kernels -- on every channel (cfg2, cfg3) or on
       a spread of channels (cfg4, cfg5).
cfg2 at full size is additionally compared with the C oracle itself.

Synthetic inputs follow SURVEY.md section 8(d): shifts (k mod s, (k div s) mod s),
lambda 0.01, BTV(3, 0.5), blur (3, 1.0) only where the configuration names one
(cfg2); the blurred variants of the other configurations are covered too.
"""
import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu

TOL = {0: 1e-12, 1: 2e-5}


from parity_log import note, relerr  # noqa: E402  (max |a - ref| / max(1, |ref|), logged when SRMAP_PARITY_LOG is set)


def cfg_shifts(K, s):
    return [[k % s, (k // s) % s] for k in range(K)]


# name: K, s, regs [(kind, lambda, range, decay)], blur of the stated configuration
CONFIGS = {
    "cfg2": dict(K=16, s=4, regs=[(2, 0.01, 3, 0.5)], blur=(3, 1.0)),
    "cfg3": dict(K=16, s=4, regs=[(2, 0.01, 3, 0.5)], blur=(0, 0.0)),
    "cfg4": dict(K=9, s=3, regs=[(0, 0.01, 0, 0.0)], blur=(0, 0.0)),
    "cfg5": dict(K=64, s=4, regs=[(2, 0.01, 3, 0.5), (1, 0.01, 0, 0.0)], blur=(0, 0.0)),
}

# reduced replicas: (config, C, LR width, LR height); HR sizes span several tiles and are not multiples of 64
REDUCED = [
    ("cfg2", 1, 70, 23),
    ("cfg3", 3, 67, 19),     # 16-frame RGB, BTV
    ("cfg4", 5, 87, 31),     # 9 frames, s = 3 -> HR 261 x 93, TV, 5 channels
    ("cfg5", 4, 66, 17),     # 64 frames, BTV(3) then 3-D TV, 4 channels
]


@pytest.fixture(scope="module")
def sr():
    import srmap
    return srmap


@pytest.fixture(scope="module")
def ctx(sr):
    return sr.Context(0)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("blurred", [False, True])
@pytest.mark.parametrize("case", range(len(REDUCED)))
def test_reduced_replica_matches_oracle(sr, ctx, case, blurred, dtype):
    name, C, w, h = REDUCED[case]
    cf = CONFIGS[name]
    K, s = cf["K"], cf["s"]
    b, sigma = (3, 1.0) if blurred else cf["blur"]
    if blurred and cf["blur"][0]:
        pytest.skip("the stated configuration is already the blurred one")
    W, H = w * s, h * s
    shifts = cfg_shifts(K, s)
    rng = np.random.default_rng(900 + case)
    model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=b, blur_sigma=sigma)
    gt = rng.random((C, H, W))
    lr = np.stack([model.apply(gt, k) for k in range(K)]) + (5 / 255) * rng.standard_normal((K, C, h, w))
    ref = orc.Problem(model, lr)
    p = sr.Problem(ctx, W, H, C, K, s, shifts, b, sigma, dtype)
    p.set_observations(lr)
    x = np.clip(gt + 0.05 * rng.standard_normal(gt.shape), 0, 1)
    x = np.round(x * 256) / 256  # ties exercise sgn(0) = 0
    for kind, lam, rg, dc in cf["regs"]:
        i = p.add_regularizer(kind, lam, rg, dc)
        j = ref.add_regularizer(kind, lam, rg, dc)
        assert i == j
        wts = 1.0 / np.maximum(1e-5, orc.reg_values(kind, x, rg, dc))  # w = 1/max(1e-5, r(x0)), SURVEY 8(d)
        wts = np.minimum(wts, 50.0)  # keep f32 within its tolerance; f64 is insensitive
        p.set_irls_weights(i, wts)
        ref.set_irls_weights(j, wts)
    f_ref, g_ref = ref.objective(x)
    tol = TOL[dtype]
    for impl in (sr.IMPL_AUTO, sr.IMPL_DIRECT):
        p.set_impl(impl)
        f, g = p.eval(x)
        assert abs(f - f_ref) <= (tol if dtype == 0 else 1e-5) * max(1.0, abs(f_ref)), (name, impl)
        assert relerr(g, g_ref) <= tol, (name, impl)
    p.set_impl(sr.IMPL_AUTO)
    fd, gd = p.eval(x, sr.TERM_DATA)
    fd_ref, gd_ref = ref.data_term(x)
    assert abs(fd - fd_ref) <= (tol if dtype == 0 else 1e-5) * max(1.0, abs(fd_ref))
    assert relerr(gd, gd_ref) <= tol


def test_cfg2_full_size_matches_oracle(sr, ctx):
    """cfg2 at its full size (2048 x 2048, 16 frames, blur 3 + BTV 3): one f64 evaluation against the C oracle."""
    cf = CONFIGS["cfg2"]
    K, s, w, h = cf["K"], cf["s"], 512, 512
    W, H = w * s, h * s
    shifts = cfg_shifts(K, s)
    rng = np.random.default_rng(2048)
    lr = rng.random((K, 1, h, w))
    x = rng.random((1, H, W))
    wts = 0.5 + rng.random((1, H, W))
    model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=3, blur_sigma=1.0)
    ref = orc.Problem(model, lr)
    ref.add_regularizer(2, 0.01, 3, 0.5)
    ref.set_irls_weights(0, wts)
    f_ref, g_ref = ref.objective(x)
    p = sr.Problem(ctx, W, H, 1, K, s, shifts, 3, 1.0, sr.F64)
    p.set_observations(lr)
    p.set_irls_weights(p.add_regularizer(sr.REG_BTV, 0.01, 3, 0.5), wts)
    f, g = p.eval(x)
    assert abs(f - f_ref) <= 1e-12 * abs(f_ref)
    assert relerr(g, g_ref) <= 1e-12


# ------------------------------------------------------------------ torch restatement (whole-array, on the GPU)
def t_shift(img, dx, dy):
    """out(r, c) = img(r - dy, c - dx), zeros outside (integer MotionModule, motion_module.cpp:18-38)."""
    import torch
    out = torch.zeros_like(img)
    H, W = img.shape[-2:]
    r0, r1 = max(0, dy), min(H, H + dy)
    c0, c1 = max(0, dx), min(W, W + dx)
    if r0 < r1 and c0 < c1:
        out[..., r0:r1, c0:c1] = img[..., r0 - dy:r1 - dy, c0 - dx:c1 - dx]
    return out


def t_blur(img, k2):
    """zero-padded correlation with the b x b kernel (filter2D BORDER_CONSTANT, matrix_util.cpp:12-29)."""
    import torch
    b = len(k2)
    if b == 1:
        return img.clone()
    h = b // 2
    out = torch.zeros_like(img)
    for a in range(b):
        for e in range(b):
            out += float(k2[a][e]) * t_shift(img, -(e - h), -(a - h))
    return out


def t_data_term(x, y, shifts, s, k2):
    """cost = s^2 sum_k ||A_k x - y_k||^2, g = 2 s^2 sum_k A_k^T (A_k x - y_k); every stage clipped to H x W."""
    import torch
    g = torch.zeros_like(x)
    cost = 0.0
    k2t = [[k2[e][a] for e in range(len(k2))] for a in range(len(k2))]
    for k, (dx, dy) in enumerate(shifts):
        r = t_blur(t_shift(x, dx, dy), k2)[..., ::s, ::s] - y[k]
        cost += float((r * r).sum())
        u = torch.zeros_like(x)
        u[..., ::s, ::s] = r
        g += t_shift(t_blur(u, k2t), -dx, -dy)
    return (s * s) * cost, (2.0 * s * s) * g


def t_diff(x, i, j):
    """d(p) = x(p) - x(p + (i, j)) where the neighbour exists, else 0."""
    import torch
    H, W = x.shape[-2:]
    d = torch.zeros_like(x)
    d[..., :H - i, :W - j] = x[..., :H - i, :W - j] - x[..., i:, j:]
    return d


def t_btv(x, lam_w, R, alpha):
    import torch
    H, W = x.shape[-2:]
    r = torch.zeros_like(x)
    for i in range(R + 1):
        for j in range(R + 1):
            r += (alpha ** (i + j)) * t_diff(x, i, j).abs()
    cr = 2.0 * lam_w * r
    didi = torch.zeros_like(x)
    for i in range(R):
        for j in range(R):
            didi += (alpha ** (i + j)) * torch.sign(t_diff(x, i, j))
    g = cr * didi
    src = cr.clone()
    src[..., 0, 0] = 0.0
    for i in range(R):
        for j in range(R):
            if i == 0 and j == 0:
                continue
            g[..., i:, j:] += src[..., :H - i, :W - j] * (-torch.sign(t_diff(x, i, j)[..., :H - i, :W - j])) * (alpha ** (i + j))
    return r, g


def t_tv(x, lam_w, xn=None, xp=None, lam_w_prev=None):
    """2-D TV of the planes x [c][H][W]; 3-D when xn (planes c+1, zeros-masked by the caller through has_next)
    and xp are given: xn / xp are (tensor, mask) pairs."""
    import torch
    dx = -t_diff(x, 0, 1)
    dy = -t_diff(x, 1, 0)
    r = dy.abs() + dx.abs()
    if xn is not None:
        nxt, has_next = xn
        r = r + has_next * (nxt - x).abs()
    cr = 2.0 * lam_w * r
    g = cr * (-torch.sign(dx) - torch.sign(dy))
    g[..., :, 1:] += cr[..., :, :-1] * torch.sign(dx[..., :, :-1])
    g[..., 1:, :] += cr[..., :-1, :] * torch.sign(dy[..., :-1, :])
    if xp is not None:
        prv, has_prev = xp
        # r of the previous channel at the same pixel: |dy| + |dx| of that plane + |x - prv|
        rp = t_diff(prv, 1, 0).abs() + t_diff(prv, 0, 1).abs() + (x - prv).abs()
        g = g + has_prev * (2.0 * lam_w_prev * rp) * torch.sign(x - prv)
    return r, g


def _full_size_case(sr, ctx, name, W, H, C, check_channels, blurred=False):
    import torch
    cf = CONFIGS[name]
    K, s = cf["K"], cf["s"]
    b, sigma = (3, 1.0) if blurred else cf["blur"]
    w, h = W // s, H // s
    shifts = cfg_shifts(K, s)
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(len(name) * 1000 + W)
    y = torch.rand((K, C, h, w), generator=gen, device=dev, dtype=torch.float64)
    x = torch.rand((C, H, W), generator=gen, device=dev, dtype=torch.float64)
    x = torch.round(x * 1024) / 1024  # ties
    torch.cuda.synchronize()  # the library enqueues on its own stream
    p = sr.Problem(ctx, W, H, C, K, s, shifts, b, sigma, sr.F64)
    p.set_observations_device(y.data_ptr())
    for kind, lam, rg, dc in cf["regs"]:
        i = p.add_regularizer(kind, lam, rg, dc)
        p.update_irls_weights_device(i, x.data_ptr())  # w = 1/max(1e-5, r(x0)) on device
    g_t = torch.empty_like(x)
    g_d = torch.empty_like(x)
    p.set_impl(sr.IMPL_AUTO)
    f_t = p.eval_device(x.data_ptr(), g_t.data_ptr(), sr.TERM_ALL, want_cost=True)
    p.set_impl(sr.IMPL_DIRECT)
    f_d = p.eval_device(x.data_ptr(), g_d.data_ptr(), sr.TERM_ALL, want_cost=True)
    torch.cuda.synchronize()
    assert note(abs(f_t - f_d) / abs(f_d), "cost tiled-vs-direct") <= 1e-12, (f_t, f_d)
    den = torch.clamp(g_d.abs(), min=1.0)
    assert note(float(((g_t - g_d).abs() / den).max()), "grad tiled-vs-direct") <= 1e-12
    # torch restatement on the selected channels
    k1, k2 = (orc.gaussian_kernel(b, sigma) if b else (None, [[1.0]]))
    k2 = [[float(v) for v in row] for row in np.asarray(k2)]
    total = 0.0
    for c in check_channels:
        xc = x[c:c + 1]
        fc, gc = t_data_term(xc, y[:, c:c + 1], shifts, s, k2)
        for kind, lam, rg, dc in cf["regs"]:
            if kind == 2:
                r = torch.zeros_like(xc)
                for i in range(rg + 1):
                    for j in range(rg + 1):
                        r += (dc ** (i + j)) * t_diff(xc, i, j).abs()
                wts = 1.0 / torch.clamp(r, min=1e-5)
                r2, gr = t_btv(xc, lam * wts, rg, dc)
            else:
                d3 = kind == 1
                has_next = float(d3 and c + 1 < C)
                has_prev = float(d3 and c > 0)
                nxt = x[c + 1:c + 2] if has_next else xc
                prv = x[c - 1:c] if has_prev else xc
                r = t_diff(xc, 1, 0).abs() + t_diff(xc, 0, 1).abs() + has_next * (nxt - xc).abs()
                wts = 1.0 / torch.clamp(r, min=1e-5)
                rp = t_diff(prv, 1, 0).abs() + t_diff(prv, 0, 1).abs() + (xc - prv).abs()
                wp = 1.0 / torch.clamp(rp, min=1e-5)
                r2, gr = t_tv(xc, lam * wts, (nxt, has_next) if d3 else None, (prv, has_prev) if d3 else None, lam * wp)
            fc += float((lam * wts * r2 * r2).sum())
            gc = gc + gr
        total += fc
        den = torch.clamp(gc.abs(), min=1.0)
        err = note(float(((g_t[c:c + 1] - gc).abs() / den).max()), "grad vs torch")
        assert err <= 1e-12, (name, c, err)   # measured <= 5e-15 (profiles/r06_parity_errors.txt)
    if len(check_channels) == C:
        assert note(abs(total - f_t) / abs(f_t), "cost vs torch") <= 1e-12, (total, f_t)
    del p
    torch.cuda.empty_cache()


def test_cfg2_full_size_tiled_direct_torch(sr, ctx):
    _full_size_case(sr, ctx, "cfg2", 2048, 2048, 1, [0])


@pytest.mark.parametrize("blurred", [False, True])
def test_cfg3_full_size(sr, ctx, blurred):
    """16-frame RGB, 4x -> 4096 x 4096, BTV: all three channels."""
    _full_size_case(sr, ctx, "cfg3", 4096, 4096, 3, [0, 1, 2], blurred)


def test_cfg4_full_size(sr, ctx):
    """9 frames x 128 channels, 3x -> 1023 x 1023 (HR not a multiple of the tile), TV."""
    _full_size_case(sr, ctx, "cfg4", 1023, 1023, 128, [0, 63, 127])


def test_cfg5_full_size(sr, ctx):
    """64 frames x 256 channels, 4x -> 2048 x 2048, BTV(3) + 3-D TV (previous / next channel coupling)."""
    _full_size_case(sr, ctx, "cfg5", 2048, 2048, 256, [0, 1, 128, 255])


def test_many_partials_one_launch_finish_and_gd(sr, ctx):
    """More than 16 384 cost partials in ONE launch (70 channels x 256 tile rows = 17 920 tiles + border blocks; BASELINE
    configs[2] has 24 576): the tile kernel's last workgroup still reduces them (several polling chunks) and hands the
    solver g.d from the same launch.  A short CG run on the tile path must follow the run on the direct kernels (cost and
    g.d from separate reduction launches there) evaluation for evaluation."""
    rng = np.random.default_rng(31)
    s, K, C, h, w = 4, 4, 70, 512, 64
    H, W = h * s, w * s
    shifts = [[0, 0], [-1, -2], [-3, -1], [-2, -3]]
    lr = rng.random((K, C, h, w))
    x0 = rng.random((C, H, W))
    res = {}
    for name, impl in (("tiled", sr.IMPL_TILED), ("direct", sr.IMPL_DIRECT)):
        p = sr.Problem(ctx, W, H, C, K, s, shifts, 3, 1.0, sr.F64)
        p.set_impl(impl)
        p.set_observations(lr)
        p.add_regularizer(sr.REG_BTV, 0.01, 3, 0.5)
        f, g = p.eval(x0)
        x, its, nfev, term, trace = p.cg_trace(x0, 0.0, 0.0, 0.0, 3)
        res[name] = (f, g, x, its, nfev, term, trace)
        del p
    (f1, g1, x1, i1, n1, t1, tr1), (f2, g2, x2, i2, n2, t2, tr2) = res["tiled"], res["direct"]
    assert abs(f1 - f2) <= 1e-12 * abs(f2) and relerr(g1, g2) <= 1e-12
    assert (i1, n1, t1) == (i2, n2, t2) and len(tr1) == len(tr2)
    assert np.max(np.abs(tr1 - tr2) / np.maximum(1.0, np.abs(tr2))) <= 1e-11
    assert np.max(np.abs(x1 - x2)) <= 1e-9


def test_indexing_beyond_2_31_elements(sr):
    """a15 (GetPixelIndex, 64-bit offsets): K * C * n = 2.28 G observation elements (> 2^31) at cfg5-like extents;
    the gradient of the first / middle / LAST channel of the many-channel problem must be bit-identical to that of a
    single-channel problem built from the same slices.  Device-generated data (no multi-GB host arrays), f32."""
    import torch
    C, K, s, W = 136, 64, 4, 2048
    H, w, h = W, W // s, W // s
    dev = torch.device("cuda", 0)
    shifts = [[k % s, (k // s) % s] for k in range(K)]
    ctx = sr.Context(0)
    big = sr.Problem(ctx, W, H, C, K, s, shifts, 3, 1.0, sr.F32)
    g = torch.Generator(device=dev); g.manual_seed(1)
    y = torch.rand((K, C, h, w), generator=g, device=dev, dtype=torch.float32)
    assert y.numel() > 2 ** 31
    big.set_observations_device(y.data_ptr())
    big.add_regularizer(sr.REG_BTV, 0.01, 3, 0.5)
    x = torch.rand((C, H, W), generator=g, device=dev, dtype=torch.float32)
    gbig = torch.empty_like(x)
    torch.cuda.synchronize()  # the library enqueues on its own (non-blocking) stream: torch's generators must be done
    cost_big = big.eval_device(x.data_ptr(), gbig.data_ptr(), sr.TERM_ALL, want_cost=True)
    torch.cuda.synchronize()
    assert np.isfinite(cost_big) and cost_big > 0
    for c in (0, C // 2, C - 1):
        one = sr.Problem(ctx, W, H, 1, K, s, shifts, 3, 1.0, sr.F32)
        yc = y[:, c:c + 1].contiguous()
        one.set_observations_device(yc.data_ptr())
        one.add_regularizer(sr.REG_BTV, 0.01, 3, 0.5)
        xc = x[c:c + 1].contiguous()
        gc = torch.empty_like(xc)
        torch.cuda.synchronize()
        one.eval_device(xc.data_ptr(), gc.data_ptr(), sr.TERM_ALL, want_cost=True)
        torch.cuda.synchronize()
        assert torch.equal(gc[0], gbig[c]), "channel %d differs" % c
        del one
    del big, x, y, gbig
    torch.cuda.empty_cache()
