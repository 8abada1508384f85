"""Independent numpy/scipy restatement of the forward model (the dense-operator
construction the reference itself uses as its oracle: MotionModule /
DownsamplingModule::GetOperatorMatrix, ConvertKernelToOperatorMatrix,
ImageModel::GetModelMatrix -- motion_module.cpp:53-73,
downsampling_module.cpp:41-64, degradation_operator.cpp:21-74,
image_model.cpp:103-118) against the C oracle, plus algebraic identities the
GPU path relies on (SURVEY.md section 8 a')."""
import numpy as np
import pytest
from scipy import ndimage

import oracle as orc


def dense_motion(W, H, dx, dy):
    M = np.zeros((W * H, W * H))
    for r in range(H):
        for c in range(W):
            sr, sc = r - int(dy), c - int(dx)
            if 0 <= sr < H and 0 <= sc < W:
                M[r * W + c, sr * W + sc] = 1
    return M


def dense_blur(W, H, k2):
    b = k2.shape[0]
    h = b // 2
    B = np.zeros((W * H, W * H))
    for r in range(H):
        for c in range(W):
            for a in range(b):
                for e in range(b):
                    rr, cc = r + a - h, c + e - h
                    if 0 <= rr < H and 0 <= cc < W:
                        B[r * W + c, rr * W + cc] += k2[a, e]
    return B


def dense_down(W, H, s):
    rows = [r * W + c for r in range(0, H, s) for c in range(0, W, s)]
    D = np.zeros((len(rows), W * H))
    for i, j in enumerate(rows):
        D[i, j] = 1
    return D


@pytest.mark.parametrize("s", [2, 3])
@pytest.mark.parametrize("b", [0, 3])
def test_dense_operator_and_adjoint(s, b):
    W = H = 12
    rng = np.random.default_rng(100 * s + b)
    x = rng.standard_normal((1, H, W))
    y = rng.standard_normal((1, H // s, W // s))
    shifts = [[dx, dy] for dx in range(-2, 3) for dy in range(-2, 3)]
    model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=b, blur_sigma=1.0)
    k2 = orc.gaussian_kernel(b, 1.0)[1] if b else np.ones((1, 1))
    for k, (dx, dy) in enumerate(shifts):
        A = dense_down(W, H, s) @ dense_blur(W, H, k2) @ dense_motion(W, H, dx, dy)
        assert np.allclose(model.apply(x, k).ravel(), A @ x.ravel(), rtol=0, atol=1e-13)
        assert np.allclose(model.apply_transpose(y, k).ravel(), A.T @ y.ravel(), rtol=0, atol=1e-13)


def test_scipy_blur_and_shift():
    rng = np.random.default_rng(1)
    img = rng.random((17, 13))
    for b, sig in ((3, 1.0), (5, 1.3), (7, 2.0)):
        k2 = orc.gaussian_kernel(b, sig)[1]
        assert np.allclose(orc.filter2d(img, k2), ndimage.correlate(img, k2, mode="constant", cval=0.0), atol=1e-14)
        x = np.arange(b) - (b - 1) / 2
        g = np.exp(-x * x / (2 * sig * sig))
        assert np.allclose(orc.gaussian_kernel(b, sig)[0], g / g.sum(), rtol=1e-15)
    for dx, dy in ((0, 0), (2, -1), (-3, 4), (20, 0)):
        ref = np.zeros_like(img)
        H, W = img.shape
        for r in range(H):
            for c in range(W):
                if 0 <= r - dy < H and 0 <= c - dx < W:
                    ref[r, c] = img[r - dy, c - dx]
        assert np.array_equal(orc.warp_shift(img, dx, dy), ref)


def test_fractional_warp_is_quantised_bilinear():
    """Sub-pixel warpAffine restatement (PARITY UNPINNED by the reference):
    coordinates quantised to 1/32 px, bilinear, zero outside."""
    rng = np.random.default_rng(2)
    img = rng.random((9, 11))
    H, W = img.shape
    for dx, dy in ((0.5, 0.25), (-1.3, 2.71), (0.01, -0.99)):
        qx, qy = np.round(-dx * 32) / 32, np.round(-dy * 32) / 32
        out = orc.warp_shift(img, dx, dy)
        pad = np.zeros((H + 8, W + 8))
        pad[4:-4, 4:-4] = img
        for r in range(H):
            for c in range(W):
                sx, sy = c + qx, r + qy
                ix, iy = int(np.floor(sx)), int(np.floor(sy))
                fx, fy = sx - ix, sy - iy
                v = 0.0
                if -4 <= ix < W + 3 and -4 <= iy < H + 3:
                    p = pad[iy + 4:iy + 6, ix + 4:ix + 6]
                    v = p[0, 0] * (1 - fx) * (1 - fy) + p[0, 1] * fx * (1 - fy) + p[1, 0] * (1 - fx) * fy + p[1, 1] * fx * fy
                assert abs(out[r, c] - v) < 1e-14


@pytest.mark.parametrize("s", [2, 3, 4, 5, 7])
def test_nearest_maps_are_regular(s):
    """resize(INTER_NEAREST) maps on the solver path: decimation = s*j, NN
    upsampling = floor(x/s) (double arithmetic of cv::resize included)."""
    for n in (1, 2, 5, 64, 341, 512, 1000):
        assert orc.downsampled_len(n * s, s) == n
        assert np.array_equal(orc.nearest_map(n * s, n), np.arange(n) * s)
        assert np.array_equal(orc.nearest_map(n, n * s), np.arange(n * s) // s)


def test_data_term_equals_lr_form():
    """cost = s^2 * sum ||A_k x - y_k||^2, grad = 2 s^2 sum A_k^T r_k."""
    rng = np.random.default_rng(9)
    s, C, h, w = 3, 2, 5, 4
    shifts = [[0, 0], [1, 2], [-1, 1], [2, -2]]
    model = orc.ImageModel(scale=s, shifts=shifts, blur_ksize=3, blur_sigma=0.8)
    lr = rng.random((4, C, h, w))
    x = rng.random((C, h * s, w * s))
    prob = orc.Problem(model, lr)
    f, g = prob.data_term(x)
    f2, g2 = 0.0, np.zeros_like(x)
    for k in range(4):
        r = model.apply(x, k) - lr[k]
        f2 += s * s * (r ** 2).sum()
        g2 += 2 * s * s * model.apply_transpose(r, k)
    assert f == pytest.approx(f2, rel=1e-13)
    assert np.allclose(g.reshape(x.shape), g2, rtol=1e-13, atol=1e-14)


def test_objective_gradient_matches_fd_for_data_term():
    rng = np.random.default_rng(4)
    model = orc.ImageModel(scale=2, shifts=[[0, 0], [1, 1], [0, 1]], blur_ksize=3, blur_sigma=1.0)
    lr = rng.random((3, 1, 4, 4))
    prob = orc.Problem(model, lr)
    x = rng.random(64)
    f, g = prob.objective(x)
    for i in (0, 7, 27, 63):
        e = np.zeros(64); e[i] = 1e-6
        fd = (prob.objective(x + e, False)[0] - prob.objective(x - e, False)[0]) / 2e-6
        assert abs(fd - g[i]) < 1e-6


# ---------------------------------------------------------------------------
# Independent second restatement of the regulariser gradients.  The C oracle
# (oracle/srmap_oracle.c) walks pixels like the reference does; the functions
# below are written from the formulas as whole-array
# shifted differences, so a transcription slip in the pixel loops cannot be
# shared by both.  Reference: btv_regularizer.cpp:93-170, tv_regularizer.cpp:135-227.
def _sgn(a):
    return np.sign(a)  # sgn(0) = 0


def np_btv(x, c, R, alpha, skip_origin=True):
    """values r and gradient for constants c; x, c: [C][H][W]."""
    Cn, H, W = x.shape
    r = np.zeros_like(x)
    for i in range(R + 1):              # inclusive window for the values
        for j in range(R + 1):
            d = np.zeros_like(x)
            d[:, :H - i, :W - j] = x[:, :H - i, :W - j] - x[:, i:, j:]
            r += alpha ** (i + j) * np.abs(d)
    cr = 2.0 * c * r
    g = np.zeros_like(x)
    didi = np.zeros_like(x)
    for i in range(R):                  # exclusive window in the gradient
        for j in range(R):
            d = np.zeros_like(x)
            d[:, :H - i, :W - j] = x[:, :H - i, :W - j] - x[:, i:, j:]
            didi += alpha ** (i + j) * _sgn(d)
    g += cr * didi
    src = cr.copy()
    if skip_origin:
        src[:, 0, 0] = 0.0              # the absolute pixel (0,0) never back-propagates
    for i in range(R):
        for j in range(R):
            # p = q + (i, j):  g[p] += 2 c[q] r[q] * (-sgn(x[q] - x[p])) * alpha^(i+j)
            t = np.zeros_like(x)
            t[:, i:, j:] = src[:, :H - i, :W - j] * (-_sgn(x[:, :H - i, :W - j] - x[:, i:, j:])) * alpha ** (i + j)
            if i == 0 and j == 0:
                # q == p: the reference still executes this tap (diff = 0 -> didj = 0), except at (0,0)
                t[:] = 0.0
            g += t
    return r, g


def np_tv(x, c, use3d):
    Cn, H, W = x.shape
    dx = np.zeros_like(x); dx[:, :, :-1] = x[:, :, 1:] - x[:, :, :-1]
    dy = np.zeros_like(x); dy[:, :-1, :] = x[:, 1:, :] - x[:, :-1, :]
    dz = np.zeros_like(x)
    if use3d and Cn > 1:
        dz[:-1] = x[1:] - x[:-1]
    r = np.abs(dy) + np.abs(dx) + (np.abs(dz) if use3d else 0.0)
    cr = 2.0 * c * r
    g = cr * (-_sgn(dx) - _sgn(dy))     # no z self term (the reference's omission)
    g[:, :, 1:] += cr[:, :, :-1] * _sgn(dx[:, :, :-1])
    g[:, 1:, :] += cr[:, :-1, :] * _sgn(dy[:, :-1, :])
    if use3d and Cn > 1:
        g[1:] += cr[:-1] * _sgn(dz[:-1])
    return r, g


@pytest.mark.parametrize("shape", [(1, 5, 5), (3, 13, 17), (2, 1, 9), (2, 8, 1), (4, 24, 31)])
@pytest.mark.parametrize("R,alpha", [(1, 0.25), (2, 0.5), (3, 0.5), (3, 1.0), (4, 0.7)])
def test_btv_gradient_against_independent_numpy(shape, R, alpha):
    rng = np.random.default_rng(R * 1000 + shape[1] * shape[2])
    x = np.round(rng.random(shape) * 8) / 8   # ties exercise sgn(0) = 0
    c = 0.25 + rng.random(shape)
    r_ref, g_ref = np_btv(x, c, R, alpha)
    r, g = orc.reg_values_and_gradient(orc.REG_BTV, x, c, R, alpha)
    assert np.max(np.abs(r - r_ref)) <= 1e-13
    assert np.max(np.abs(g - g_ref)) <= 1e-12 * max(1.0, np.max(np.abs(g_ref)))
    if R == 1:
        assert np.all(g == 0)
    # the (0,0) exception really is in force: a restatement without it differs at the origin's neighbours
    if R > 1 and shape[1] > 1 and shape[2] > 1:
        _, g_no = np_btv(x, c, R, alpha, skip_origin=False)
        moved = np.abs(g_no - g_ref) > 0
        assert not moved[:, R:, :].any() and not moved[:, :, R:].any()
        if np.any(x[:, 0, 0] != x[:, 0, 1]):
            assert moved[:, 0, 1].any() and np.max(np.abs(g - g_no)) > 0


@pytest.mark.parametrize("shape", [(1, 3, 3), (3, 13, 17), (2, 1, 9), (5, 8, 1), (4, 24, 31)])
@pytest.mark.parametrize("use3d", [False, True])
def test_tv_gradient_against_independent_numpy(shape, use3d):
    rng = np.random.default_rng(17 + shape[0] * shape[1] * shape[2] + int(use3d))
    x = np.round(rng.random(shape) * 8) / 8
    c = 0.25 + rng.random(shape)
    r_ref, g_ref = np_tv(x, c, use3d)
    r, g = orc.reg_values_and_gradient(orc.REG_TV3D if use3d else orc.REG_TV, x, c)
    assert np.max(np.abs(r - r_ref)) <= 1e-13
    assert np.max(np.abs(g - g_ref)) <= 1e-12 * max(1.0, np.max(np.abs(g_ref)))


def test_tv_gradient_reference_probe_value():
    """The one gradient the survey's compiled-reference probe recorded: TV gradient of the
    3x3 test image with unit constants = 0 -8 0 0 12 18 -12 -6 -4."""
    import json, os
    lit = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_literals.json")))
    img = np.array(lit["tv"]["image"], dtype=float).reshape(1, 3, 3)
    r, g = orc.reg_values_and_gradient(orc.REG_TV, img, np.ones_like(img))
    assert np.array_equal(r.ravel(), np.array(lit["tv"]["expected"], dtype=float))
    assert np.array_equal(g.ravel(), np.array([0, -8, 0, 0, 12, 18, -12, -6, -4], dtype=float))
    r2, g2 = np_tv(img, np.ones_like(img), False)
    assert np.array_equal(g2.ravel(), g.ravel())
