"""registration::TranslationalRegistration on the GPU (srmap_register_translational).

The reference's algorithm is an OpenCV feature pipeline (registration.cpp:41-157) that cannot be restated here;
the contract tested is the reference's own (test/test_registration.cpp:27-68): shifts applied with MotionModule
are recovered to kTranslationEstimateErrorTolerance = 0.01 px.  Shifted frames come from the library's own
MotionModule (srmap_apply at scale 1 = warpAffine with zero border), which the parity tests pin against the oracle.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sr():
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "super-resolution_amd", "python"))
    import srmap
    return srmap


@pytest.fixture(scope="module")
def ctx(sr):
    return sr.Context(0)


def texture(rng, H, W):
    """Band-limited random texture + sinusoids (the reference's test image is a JPEG photograph)."""
    coarse = rng.random((H // 8 + 2, W // 8 + 2))
    r = np.arange(H) / 8.0
    c = np.arange(W) / 8.0
    r0, c0 = r.astype(int), c.astype(int)
    a, b = (c - c0)[None, :], (r - r0)[:, None]
    g = (1 - b) * ((1 - a) * coarse[r0][:, c0] + a * coarse[r0][:, c0 + 1]) + b * ((1 - a) * coarse[r0 + 1][:, c0] + a * coarse[r0 + 1][:, c0 + 1])
    yy, xx = np.mgrid[0:H, 0:W]
    return 0.6 * g + 0.2 + 0.1 * np.sin(0.21 * xx) * np.cos(0.17 * yy)


def shifted_stack(sr, ctx, img, shifts):
    H, W = img.shape
    p = sr.Problem(ctx, W, H, 1, len(shifts), 1, shifts, 0, 0.0, sr.F64)  # MotionModule only
    return np.stack([p.apply(img[None], k)[0] for k in range(len(shifts))])


@pytest.mark.parametrize("size", [(240, 320), (515, 389), (1024, 1024)])
def test_reference_test_shifts(sr, ctx, size):
    """The five shifts of test/test_registration.cpp:28-34, tolerance 0.01 px."""
    H, W = size
    rng = np.random.default_rng(H + W)
    img = texture(rng, H, W)
    truth = [[0, 0], [0, 1], [2, 0], [5, 5], [-5, -1]]
    got = ctx.register_translational(shifted_stack(sr, ctx, img, truth))
    assert got.shape == (5, 2)
    assert np.all(got[0] == 0)
    assert np.max(np.abs(got - np.array(truth, dtype=float))) <= 0.01


def test_large_and_subpixel_shifts(sr, ctx):
    rng = np.random.default_rng(3)
    img = texture(rng, 480, 640)
    truth = [[0, 0], [37, -22], [-64, 48], [0.5, -0.25], [1.75, 2.5], [-3.125, 0.875], [10.5, -7.25]]
    got = ctx.register_translational(shifted_stack(sr, ctx, img, truth))
    err = np.abs(got - np.array(truth, dtype=float))
    assert np.max(err[:3]) <= 0.01       # integer shifts: exact up to the refinement's stopping rule
    assert np.max(err[3:]) <= 0.05       # bilinear-warped content


def test_noise_and_edge_cases(sr, ctx):
    rng = np.random.default_rng(5)
    img = texture(rng, 256, 256)
    truth = [[0, 0], [3, -2], [-1.5, 4.25]]
    stack = shifted_stack(sr, ctx, img, truth) + 0.01 * rng.standard_normal((3, 256, 256))
    got = ctx.register_translational(stack)
    assert np.max(np.abs(got - np.array(truth, dtype=float))) <= 0.1
    # one image: (0, 0) (registration.cpp:170-172); none: empty sequence (:165-168)
    assert np.array_equal(ctx.register_translational(stack[:1]), np.zeros((1, 2)))
    assert ctx.register_translational(np.zeros((0, 16, 16))).shape == (0, 2)
    # featureless frames: no shift can be determined -> the integer search still answers (all candidates tie at 0
    # error), the refinement has no texture and leaves it: (0, 0) is among the ties and the first minimum wins
    flat = np.full((2, 64, 64), 0.5)
    out = ctx.register_translational(flat)
    assert np.all(np.isfinite(out))
    with pytest.raises(sr.SrmapError) as e:
        ctx.register_translational(np.zeros((2, 4, 4)))  # smaller than the estimator's 8 x 8 minimum
    assert e.value.status == sr.EINVAL


def _rotate(img, deg):
    """img rotated about its centre by `deg` degrees (bilinear, edge-clamped): motion that is NOT a translation."""
    H, W = img.shape
    yy, xx = np.mgrid[0:H, 0:W].astype(float)
    t = np.deg2rad(deg)
    cx, cy = (W - 1) / 2, (H - 1) / 2
    xs = np.clip(cx + (xx - cx) * np.cos(t) - (yy - cy) * np.sin(t), 0, W - 1.001)
    ys = np.clip(cy + (xx - cx) * np.sin(t) + (yy - cy) * np.cos(t), 0, H - 1.001)
    x0, y0 = xs.astype(int), ys.astype(int)
    a, b = xs - x0, ys - y0
    return (1 - b) * ((1 - a) * img[y0, x0] + a * img[y0, x0 + 1]) + b * ((1 - a) * img[y0 + 1, x0] + a * img[y0 + 1, x0 + 1])


def test_quality_under_noise_rotation_and_periodic_texture(sr, ctx):
    """The estimator assumes a pure translation (include/srmap.h): what it does outside that assumption, pinned.
    Noise + a 0.5 degree rotation: the translation is still found to a pixel and the quality shows a clear minimum but
    a large residual; a periodic texture: the separation collapses (the shift may be off by whole periods)."""
    rng = np.random.default_rng(11)
    img = texture(rng, 256, 320)
    truth = [[0, 0], [4, -3], [4, -3]]
    stack = shifted_stack(sr, ctx, img, truth)
    clean = stack.copy()
    stack[2] = _rotate(stack[2], 0.5)
    stack += 0.02 * rng.standard_normal(stack.shape)
    got, q = ctx.register_translational(stack, with_quality=True)
    print(got, q)
    assert np.max(np.abs(got[1] - truth[1])) <= 0.1 and np.max(np.abs(got[2] - truth[2])) <= 1.0
    assert q[1, 0] > 0.5 and q[2, 0] > 0.5                 # one clear minimum in both
    assert q[1, 1] < 0.04 and q[2, 1] > q[1, 1]            # residual: noise only vs noise + rotation
    _, q_clean = ctx.register_translational(clean, with_quality=True)
    assert q_clean[1, 1] < 1e-6 and q_clean[1, 0] > 0.9    # exact translation: zero residual (the coarse level sees 4 / 4 and -3 / 4 px)
    # periodic texture (period 16 px in x): candidates a period apart tie
    yy, xx = np.mgrid[0:256, 0:320]
    per = 0.5 + 0.4 * np.sin(2 * np.pi * xx / 16.0) * np.sin(2 * np.pi * yy / 16.0)
    pst = shifted_stack(sr, ctx, per, [[0, 0], [3, 2]])[:, 32:-32, 32:-32].copy()  # crop: no zero border to anchor on
    got_p, q_p = ctx.register_translational(pst, with_quality=True)
    print(got_p, q_p)
    assert q_p[1, 0] < 0.2                                  # ambiguous, and reported as such
    # what it returns is an exact alias of the true shift (sin x sin repeats every half period along the diagonal):
    # zero residual, (3, 2) plus multiples of 8 of equal parity -- here (-37, 42)
    k = (got_p[1] - [3, 2]) / 8.0
    assert q_p[1, 1] < 1e-9 and np.allclose(k, np.round(k), atol=0.02) and (int(round(k[0])) + int(round(k[1]))) % 2 == 0
