"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg -- never by the product path
(``super-resolution_amd``).  See ``oracle/srmap_oracle.h`` for what each entry
point restates (reference file:line) and for the pinning status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libalglib_ref.so")

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)

FG_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, c_double_p, c_double_p)
REP_FN = C.CFUNCTYPE(None, C.c_void_p, c_double_p, C.c_double)

REG_TV, REG_TV3D, REG_BTV = 0, 1, 2


class Model(C.Structure):
    _fields_ = [("scale", C.c_int), ("num_frames", C.c_int),
                ("shifts", c_double_p), ("blur_ksize", C.c_int),
                ("blur_sigma", C.c_double)]


class Regularizer(C.Structure):
    _fields_ = [("kind", C.c_int), ("btv_range", C.c_int),
                ("btv_decay", C.c_double)]


class CgReport(C.Structure):
    _fields_ = [("termination_type", C.c_int), ("iterations", C.c_int),
                ("nfev", C.c_int), ("f", C.c_double)]


class IrlsOptions(C.Structure):
    _fields_ = [("max_num_solver_iterations", C.c_int),
                ("gradient_norm_threshold", C.c_double),
                ("cost_decrease_threshold", C.c_double),
                ("parameter_variation_threshold", C.c_double),
                ("split_channels", C.c_int),
                ("max_num_irls_iterations", C.c_int),
                ("irls_cost_difference_threshold", C.c_double)]


class SolveReport(C.Structure):
    _fields_ = [("irls_rounds", C.c_int), ("cg_iterations", C.c_int),
                ("nfev", C.c_int), ("final_cost", C.c_double)]


CG_FN = C.CFUNCTYPE(None, C.c_int, c_double_p, C.c_double, C.c_double,
                    C.c_double, C.c_int, FG_FN, REP_FN, C.c_void_p,
                    C.POINTER(CgReport))


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "srmap_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if not os.path.exists(_REF) and os.path.isdir("/root/reference/libs/alglib/src"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.sro_warp_shift.argtypes = [c_double_p, c_double_p, C.c_int, C.c_int, C.c_double, C.c_double]
        L.sro_warp_tables.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, c_int_p, c_int_p]
        L.sro_gaussian_kernel.argtypes = [C.c_int, C.c_double, c_double_p, c_double_p]
        L.sro_filter2d.argtypes = [c_double_p, c_double_p, C.c_int, C.c_int, c_double_p, C.c_int, C.c_int]
        L.sro_nearest_map.argtypes = [C.c_int, C.c_int, c_int_p]
        L.sro_resize_nearest.argtypes = [c_double_p, C.c_int, C.c_int, c_double_p, C.c_int, C.c_int]
        L.sro_resize_additive.argtypes = [c_double_p, C.c_int, C.c_int, c_double_p, C.c_int, C.c_int]
        L.sro_downsampled_len.argtypes = [C.c_int, C.c_int]
        L.sro_downsampled_len.restype = C.c_int
        L.sro_model_apply.argtypes = [C.POINTER(Model), C.c_int, c_double_p, C.c_int, C.c_int, C.c_int, c_double_p]
        L.sro_model_apply_transpose.argtypes = [C.POINTER(Model), C.c_int, c_double_p, C.c_int, C.c_int, C.c_int, c_double_p]
        L.sro_reg_values.argtypes = [C.POINTER(Regularizer), c_double_p, C.c_int, C.c_int, C.c_int, c_double_p]
        L.sro_reg_values_and_gradient.argtypes = [C.POINTER(Regularizer), c_double_p, c_double_p, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p]
        L.sro_problem_create.argtypes = [C.POINTER(Model), c_double_p, C.c_int, C.c_int, C.c_int]
        L.sro_problem_create.restype = C.c_void_p
        L.sro_problem_destroy.argtypes = [C.c_void_p]
        L.sro_problem_add_regularizer.argtypes = [C.c_void_p, C.POINTER(Regularizer), C.c_double]
        L.sro_problem_add_regularizer.restype = C.c_int
        L.sro_problem_set_irls_weights.argtypes = [C.c_void_p, C.c_int, c_double_p]
        L.sro_data_term.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.sro_data_term.restype = C.c_double
        L.sro_irls_reg_term.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p]
        L.sro_irls_reg_term.restype = C.c_double
        L.sro_objective.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.sro_objective.restype = C.c_double
        L.sro_mincg.argtypes = [C.c_int, c_double_p, C.c_double, C.c_double, C.c_double, C.c_int, FG_FN, REP_FN, C.c_void_p, C.POINTER(CgReport)]
        L.sro_irls_options_default.argtypes = [C.POINTER(IrlsOptions)]
        L.sro_irls_solve.argtypes = [C.c_void_p, C.POINTER(IrlsOptions), c_double_p, c_double_p, C.c_void_p, C.POINTER(SolveReport)]
        L.sro_psnr.argtypes = [c_double_p, c_double_p, C.c_long]
        L.sro_psnr.restype = C.c_double
        _lib = L
    return _lib


def have_ref():
    build()
    return os.path.exists(_REF)


def ref():
    """The reference's vendored ALGLIB (oracle/_ref), or None."""
    global _ref
    if _ref is None and have_ref():
        R = C.CDLL(_REF)
        R.ref_mincg.argtypes = [C.c_int, c_double_p, C.c_double, C.c_double, C.c_double, C.c_int, FG_FN, REP_FN, C.c_void_p, C.POINTER(CgReport)]
        _ref = R
    return _ref


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_double_p)


# ---------------------------------------------------------------- image ops
def warp_shift(img, dx, dy):
    src, ps = _d(img)
    H, W = src.shape
    dst = np.empty_like(src)
    lib().sro_warp_shift(ps, dst.ctypes.data_as(c_double_p), W, H, dx, dy)
    return dst


def warp_tables(W, H, dx, dy):
    X = np.empty(W, dtype=np.int32)
    Y = np.empty(H, dtype=np.int32)
    lib().sro_warp_tables(W, H, dx, dy, X.ctypes.data_as(c_int_p), Y.ctypes.data_as(c_int_p))
    return X, Y


def gaussian_kernel(ksize, sigma):
    k1 = np.empty(ksize)
    k2 = np.empty((ksize, ksize))
    lib().sro_gaussian_kernel(ksize, sigma, k1.ctypes.data_as(c_double_p), k2.ctypes.data_as(c_double_p))
    return k1, k2


def filter2d(img, kernel):
    src, ps = _d(img)
    k, pk = _d(kernel)
    H, W = src.shape
    dst = np.empty_like(src)
    lib().sro_filter2d(ps, dst.ctypes.data_as(c_double_p), W, H, pk, k.shape[1], k.shape[0])
    return dst


def nearest_map(src_len, dst_len):
    m = np.empty(dst_len, dtype=np.int32)
    lib().sro_nearest_map(src_len, dst_len, m.ctypes.data_as(c_int_p))
    return m


def resize_nearest(img, dw, dh):
    src, ps = _d(img)
    dst = np.empty((dh, dw))
    lib().sro_resize_nearest(ps, src.shape[1], src.shape[0], dst.ctypes.data_as(c_double_p), dw, dh)
    return dst


def resize_additive(img, dw, dh):
    src, ps = _d(img)
    dst = np.empty((dh, dw))
    lib().sro_resize_additive(ps, src.shape[1], src.shape[0], dst.ctypes.data_as(c_double_p), dw, dh)
    return dst


def downsampled_len(n, scale):
    return lib().sro_downsampled_len(n, scale)


class ImageModel:
    """ImageModel = [MotionModule] -> [BlurModule] -> DownsamplingModule."""

    def __init__(self, scale, shifts=None, blur_ksize=0, blur_sigma=0.0, num_frames=None):
        self.scale = scale
        self._shifts = None if shifts is None else np.ascontiguousarray(shifts, dtype=np.float64).reshape(-1, 2)
        self.num_frames = len(self._shifts) if self._shifts is not None else (num_frames or 0)
        self.blur_ksize, self.blur_sigma = blur_ksize, blur_sigma

    def struct(self, for_problem=False):
        m = Model()
        m.scale = self.scale
        m.num_frames = self.num_frames if (self._shifts is not None or for_problem) else 0
        m.shifts = self._shifts.ctypes.data_as(c_double_p) if self._shifts is not None else None
        m.blur_ksize, m.blur_sigma = self.blur_ksize, self.blur_sigma
        return m

    def apply(self, hr, k):
        """hr [C][H][W] -> lr [C][h][w]."""
        x, px = _d(hr)
        Cn, H, W = x.shape
        w, h = downsampled_len(W, self.scale), downsampled_len(H, self.scale)
        lr = np.empty((Cn, h, w))
        m = self.struct()
        lib().sro_model_apply(C.byref(m), k, px, W, H, Cn, lr.ctypes.data_as(c_double_p))
        return lr

    def apply_transpose(self, lr, k):
        y, py = _d(lr)
        Cn, h, w = y.shape
        hr = np.empty((Cn, int(h * float(self.scale)), int(w * float(self.scale))))
        m = self.struct()
        lib().sro_model_apply_transpose(C.byref(m), k, py, w, h, Cn, hr.ctypes.data_as(c_double_p))
        return hr


def _reg(kind, btv_range=0, btv_decay=0.0):
    r = Regularizer()
    r.kind, r.btv_range, r.btv_decay = kind, btv_range, btv_decay
    return r


def reg_values(kind, x, btv_range=0, btv_decay=0.0):
    a, pa = _d(x)
    Cn, H, W = a.shape
    out = np.empty_like(a)
    r = _reg(kind, btv_range, btv_decay)
    lib().sro_reg_values(C.byref(r), pa, W, H, Cn, out.ctypes.data_as(c_double_p))
    return out


def reg_values_and_gradient(kind, x, gradient_constants, btv_range=0, btv_decay=0.0):
    a, pa = _d(x)
    gc, pg = _d(gradient_constants)
    Cn, H, W = a.shape
    vals = np.empty_like(a)
    grad = np.empty_like(a)
    r = _reg(kind, btv_range, btv_decay)
    lib().sro_reg_values_and_gradient(C.byref(r), pa, pg, W, H, Cn,
                                      vals.ctypes.data_as(c_double_p), grad.ctypes.data_as(c_double_p))
    return vals, grad


class Problem:
    """MapSolver state: model, observations (stored NN-upsampled), regularizers."""

    def __init__(self, model, lr_frames):
        y, py = _d(lr_frames)
        K, Cn, h, w = y.shape
        self.model = model
        if model.num_frames == 0:
            model.num_frames = K
        assert model.num_frames == K
        self.K, self.C, self.h, self.w = K, Cn, h, w
        self.H, self.W = h * model.scale, w * model.scale
        m = model.struct(for_problem=True)
        self._p = lib().sro_problem_create(C.byref(m), py, w, h, Cn)
        self.nreg = 0

    def __del__(self):
        if getattr(self, "_p", None):
            lib().sro_problem_destroy(self._p)
            self._p = None

    def add_regularizer(self, kind, lam, btv_range=0, btv_decay=0.0):
        r = _reg(kind, btv_range, btv_decay)
        idx = lib().sro_problem_add_regularizer(self._p, C.byref(r), lam)
        assert idx >= 0
        self.nreg += 1
        return idx

    def set_irls_weights(self, reg, weights):
        if weights is None:
            lib().sro_problem_set_irls_weights(self._p, reg, None)
        else:
            w, pw = _d(weights)
            assert w.size == self.C * self.H * self.W
            lib().sro_problem_set_irls_weights(self._p, reg, pw)

    def _x(self, x):
        a, pa = _d(x)
        assert a.size == self.C * self.H * self.W
        return a, pa

    def data_term(self, x, want_grad=True):
        a, pa = self._x(x)
        g = np.zeros_like(a) if want_grad else None
        f = lib().sro_data_term(self._p, pa, g.ctypes.data_as(c_double_p) if want_grad else None)
        return f, g

    def reg_term(self, reg, x, want_grad=True):
        a, pa = self._x(x)
        g = np.zeros_like(a) if want_grad else None
        f = lib().sro_irls_reg_term(self._p, reg, pa, g.ctypes.data_as(c_double_p) if want_grad else None)
        return f, g

    def objective(self, x, want_grad=True):
        a, pa = self._x(x)
        g = np.empty_like(a) if want_grad else None
        f = lib().sro_objective(self._p, pa, g.ctypes.data_as(c_double_p) if want_grad else None)
        return f, g

    def solve(self, x0, options=None, use_alglib=False):
        a, pa = self._x(x0)
        out = np.empty_like(a)
        o = default_irls_options() if options is None else options
        rep = SolveReport()
        cg = None
        if use_alglib:
            cg = C.cast(ref().ref_mincg, C.c_void_p)
        lib().sro_irls_solve(self._p, C.byref(o), pa, out.ctypes.data_as(c_double_p), cg, C.byref(rep))
        return out, rep


def default_irls_options():
    o = IrlsOptions()
    lib().sro_irls_options_default(C.byref(o))
    return o


def mincg(fun, x0, epsg=1e-6, epsf=1e-6, epsx=1e-6, maxits=50, use_alglib=False, trace=None):
    """Minimise ``fun(x) -> (f, g)`` with the mincg restatement (or, with
    ``use_alglib``, the reference's ALGLIB from oracle/_ref)."""
    x = np.array(x0, dtype=np.float64).ravel().copy()
    n = x.size

    def _fg(_ctx, px, pg):
        xv = np.ctypeslib.as_array(px, shape=(n,))
        f, g = fun(xv.copy())
        np.ctypeslib.as_array(pg, shape=(n,))[:] = g
        return float(f)

    def _rep(_ctx, px, f):
        if trace is not None:
            trace.append((np.ctypeslib.as_array(px, shape=(n,)).copy(), float(f)))

    rep = CgReport()
    fn = ref().ref_mincg if use_alglib else lib().sro_mincg
    fn(n, x.ctypes.data_as(c_double_p), epsg, epsf, epsx, maxits, FG_FN(_fg), REP_FN(_rep), None, C.byref(rep))
    return x, rep


def psnr(gt, im):
    a, pa = _d(gt)
    b, pb = _d(im)
    return lib().sro_psnr(pa, pb, a.size)
