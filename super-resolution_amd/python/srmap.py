"""ctypes binding of libsrmap.so (include/srmap.h) for the test-suite and
bench.py.  This is harness plumbing: the product is the C-ABI library and the
C++ facade in super-resolution_amd/host.  There is no CPU fallback -- importing
works anywhere (so that symbol checks can run without a GPU), but creating a
context without a HIP device raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRMAP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libsrmap.so")

OK, EINVAL, ENOMEM, EHIP, EUNSUPPORTED = 0, 1, 2, 3, 4
F64, F32 = 0, 1
REG_TV, REG_TV3D, REG_BTV = 0, 1, 2
TERM_DATA, TERM_REG, TERM_ALL = 1, 2, 3
IMPL_AUTO, IMPL_DIRECT, IMPL_TILED = 0, 1, 2

c_double_p = C.POINTER(C.c_double)


class SrmapError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("srmap status %d: %s" % (status, message))
        self.status = status


class ProblemDesc(C.Structure):
    _fields_ = [("hr_width", C.c_int), ("hr_height", C.c_int), ("channels", C.c_int),
                ("frames", C.c_int), ("scale", C.c_int), ("shifts_xy", c_double_p),
                ("blur_ksize", C.c_int), ("blur_sigma", C.c_double), ("dtype", C.c_int)]


class IrlsOptions(C.Structure):
    _fields_ = [("struct_size", C.c_int),
                ("max_num_solver_iterations", C.c_int),
                ("gradient_norm_threshold", C.c_double),
                ("cost_decrease_threshold", C.c_double),
                ("parameter_variation_threshold", C.c_double),
                ("split_channels", C.c_int),
                ("max_num_irls_iterations", C.c_int),
                ("irls_cost_difference_threshold", C.c_double),
                ("host_paced_passes", C.c_int)]


class SolveReport(C.Structure):
    _fields_ = [("irls_rounds", C.c_int), ("cg_iterations", C.c_int), ("evaluations", C.c_int),
                ("last_termination", C.c_int), ("final_cost", C.c_double), ("loop_seconds", C.c_double),
                ("wait_seconds", C.c_double), ("waits", C.c_int)]


class ShardDesc(C.Structure):
    _fields_ = [("mode", C.c_int), ("own_row0", C.c_int), ("own_row1", C.c_int),
                ("send_up_rows", C.c_int), ("send_down_rows", C.c_int),
                ("own_ch0", C.c_int), ("own_ch1", C.c_int), ("reg_rank", C.c_int),
                ("frame_groups", C.c_int), ("frame_comm", C.c_void_p)]


SHARD_NONE, SHARD_FRAMES, SHARD_ROWS, SHARD_CHANNELS, SHARD_GRID = 0, 1, 2, 3, 4
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)
HOST_SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)

# every symbol include/srmap.h declares: (name, restype, argtypes)
_SIGNATURES = [
    ("srmap_ctx_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("srmap_ctx_destroy", None, [C.c_void_p]),
    ("srmap_last_error", C.c_char_p, [C.c_void_p]),
    ("srmap_version", C.c_char_p, []),
    ("srmap_problem_create", C.c_int, [C.c_void_p, C.POINTER(ProblemDesc), C.POINTER(C.c_void_p)]),
    ("srmap_problem_set_cost_rows", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("srmap_problem_destroy", None, [C.c_void_p]),
    ("srmap_problem_set_impl", C.c_int, [C.c_void_p, C.c_int]),
    ("srmap_problem_lr_size", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("srmap_set_observations", C.c_int, [C.c_void_p, c_double_p]),
    ("srmap_problem_active_impl", C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    ("srmap_set_observations_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("srmap_add_regularizer", C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_double, C.POINTER(C.c_int)]),
    ("srmap_clear_regularizers", C.c_int, [C.c_void_p]),
    ("srmap_set_irls_weights", C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    ("srmap_update_irls_weights_device", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    ("srmap_apply", C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p]),
    ("srmap_apply_transpose", C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p]),
    ("srmap_reg_values", C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p]),
    ("srmap_reg_values_and_gradient", C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]),
    ("srmap_eval", C.c_int, [C.c_void_p, C.c_uint, c_double_p, c_double_p, c_double_p]),
    ("srmap_eval_device", C.c_int, [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, c_double_p, C.c_void_p]),
    ("srmap_last_cost", C.c_int, [C.c_void_p, c_double_p]),
    ("srmap_device_alloc", C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    ("srmap_device_free", C.c_int, [C.c_void_p, C.c_void_p]),
    ("srmap_upload", C.c_int, [C.c_void_p, c_double_p, C.c_void_p, C.c_size_t]),
    ("srmap_download", C.c_int, [C.c_void_p, C.c_void_p, c_double_p, C.c_size_t]),
    ("srmap_channel_map", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_size_t, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    ("srmap_channel_map_device", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_size_t, c_double_p, c_double_p, c_double_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("srmap_register_translational", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p]),
    ("srmap_register_translational_ex", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p]),
    ("srmap_channel_pca", C.c_int, [C.c_void_p, C.c_int, C.c_size_t, c_double_p, c_double_p, c_double_p, c_double_p]),
    ("srmap_channel_pca_device", C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, c_double_p, c_double_p, c_double_p, C.c_void_p]),
    ("srmap_synchronize", C.c_int, [C.c_void_p]),
    ("srmap_irls_options_default", None, [C.POINTER(IrlsOptions)]),
    ("srmap_solve", C.c_int, [C.c_void_p, C.POINTER(IrlsOptions), c_double_p, c_double_p, C.POINTER(SolveReport)]),
    ("srmap_problem_selfcheck", C.c_int, [C.c_void_p, c_double_p]),
    ("srmap_comm_get_unique_id", C.c_int, [C.c_void_p, C.c_char_p]),
    ("srmap_comm_create_rccl", C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    ("srmap_comm_create_host", C.c_int, [C.c_void_p, C.c_int, C.c_int, HOST_ALLREDUCE_FN, HOST_SENDRECV_FN, C.c_void_p, C.POINTER(C.c_void_p)]),
    ("srmap_comm_destroy", None, [C.c_void_p]),
    ("srmap_comm_info", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("srmap_comm_set_overlap", C.c_int, [C.c_void_p, C.c_int]),
    ("srmap_comm_describe", C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    ("srmap_comm_split", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    ("srmap_comm_allreduce", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]),
    ("srmap_eval_sharded_device", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(ShardDesc), C.c_uint, C.c_void_p, C.c_void_p, c_double_p, C.c_void_p]),
    ("srmap_solve_sharded", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(ShardDesc), C.POINTER(IrlsOptions), c_double_p, c_double_p, C.POINTER(SolveReport)]),
    ("srmap_cg_trace", C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, c_double_p, c_double_p,
                                 C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), c_double_p, C.c_int, C.POINTER(C.c_int)]),
]
EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]

_lib = None


def load():
    """dlopen libsrmap.so; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libsrmap.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                               "there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in _SIGNATURES:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_double_p)


class Context:
    def __init__(self, device=0):
        self._h = C.c_void_p()
        st = load().srmap_ctx_create(device, C.byref(self._h))
        if st != OK:
            raise SrmapError(st, "srmap_ctx_create failed (no usable HIP device %d; no CPU path exists)" % device)

    def check(self, st):
        if st != OK:
            raise SrmapError(st, load().srmap_last_error(self._h).decode())

    def synchronize(self):
        self.check(load().srmap_synchronize(self._h))

    def channel_map(self, M, x, offset_in=None, offset_out=None):
        """out[r] = sum_c M[r, c] * (x[c] - offset_in[c]) + offset_out[r] on a planar image x[C, ...] (GPU DGEMM)."""
        Mm, pM = _d(M)
        a, pa = _d(x)
        ro, ri = Mm.shape
        assert a.shape[0] == ri
        n = a.size // ri
        out = np.empty((ro,) + a.shape[1:])
        oi = _d(offset_in) if offset_in is not None else (None, None)
        oo = _d(offset_out) if offset_out is not None else (None, None)
        self.check(load().srmap_channel_map(self._h, ro, ri, n, pM, oi[1], oo[1], pa, out.ctypes.data_as(c_double_p)))
        return out

    def channel_map_device(self, M, in_ptr, out_ptr, n, offset_in=None, offset_out=None, stream=None):
        """The same on device-resident planar f64 cubes (raw device pointers, n pixels per channel)."""
        Mm, pM = _d(M)
        ro, ri = Mm.shape
        oi = _d(offset_in) if offset_in is not None else (None, None)
        oo = _d(offset_out) if offset_out is not None else (None, None)
        self.check(load().srmap_channel_map_device(self._h, ro, ri, n, pM, oi[1], oo[1], C.c_void_p(in_ptr),
                                                   C.c_void_p(out_ptr), C.c_void_p(stream) if stream else None))

    def register_translational(self, images, with_quality=False):
        """registration::TranslationalRegistration: images [n][H][W] -> shifts [n][2] (dx, dy) relative to image 0
        (with_quality: also [n][2] = separation of the coarse minimum, RMS residual -- srmap.h)."""
        a, pa = _d(images)
        n, H, W = a.shape
        out = np.zeros((n, 2))
        if not with_quality:
            self.check(load().srmap_register_translational(self._h, n, W, H, pa, out.ctypes.data_as(c_double_p)))
            return out
        q = np.zeros((n, 2))
        self.check(load().srmap_register_translational_ex(self._h, n, W, H, pa, out.ctypes.data_as(c_double_p),
                                                          q.ctypes.data_as(c_double_p)))
        return out, q

    def pca(self, samples):
        """PCA of planar samples [rows][count] on the GPU: (mean, eigenvalues descending, basis rows = eigenvectors)."""
        a, pa = _d(samples)
        rows, count = a.shape
        mean, ev, basis = np.empty(rows), np.empty(rows), np.empty((rows, rows))
        self.check(load().srmap_channel_pca(self._h, rows, count, pa, mean.ctypes.data_as(c_double_p),
                                            ev.ctypes.data_as(c_double_p), basis.ctypes.data_as(c_double_p)))
        return mean, ev, basis

    def pca_device(self, in_ptr, rows, n, first, stride, count, stream=None):
        mean, ev, basis = np.empty(rows), np.empty(rows), np.empty((rows, rows))
        self.check(load().srmap_channel_pca_device(self._h, rows, n, C.c_void_p(in_ptr), first, stride, count,
                                                   mean.ctypes.data_as(c_double_p), ev.ctypes.data_as(c_double_p),
                                                   basis.ctypes.data_as(c_double_p), C.c_void_p(stream or 0)))
        return mean, ev, basis

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:  # at interpreter exit the module globals may be gone
            _lib.srmap_ctx_destroy(self._h)
            self._h = None


def default_irls_options():
    o = IrlsOptions()
    load().srmap_irls_options_default(C.byref(o))
    return o


class Problem:
    def __init__(self, ctx, hr_width, hr_height, channels, frames, scale, shifts=None,
                 blur_ksize=0, blur_sigma=0.0, dtype=F64):
        self.ctx = ctx
        self.W, self.H, self.C, self.K, self.s = hr_width, hr_height, channels, frames, scale
        self.dtype = dtype
        d = ProblemDesc()
        d.hr_width, d.hr_height, d.channels, d.frames, d.scale = hr_width, hr_height, channels, frames, scale
        self._shifts = None
        if shifts is not None:
            self._shifts = np.ascontiguousarray(shifts, dtype=np.float64).reshape(-1, 2)
            assert len(self._shifts) == frames
            d.shifts_xy = self._shifts.ctypes.data_as(c_double_p)
        d.blur_ksize, d.blur_sigma, d.dtype = blur_ksize, blur_sigma, dtype
        self._h = C.c_void_p()
        ctx.check(load().srmap_problem_create(ctx._h, C.byref(d), C.byref(self._h)))
        lw, lh = C.c_int(), C.c_int()
        load().srmap_problem_lr_size(self._h, C.byref(lw), C.byref(lh))
        self.w, self.h = lw.value, lh.value
        self.nreg = 0

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:  # at interpreter exit the module globals may be gone
            _lib.srmap_problem_destroy(self._h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def set_impl(self, impl):
        self.ctx.check(load().srmap_problem_set_impl(self._h, impl))

    def set_cost_rows(self, hr_row0, hr_row1):
        """Row-band sharding: count only the cost terms of HR rows [hr_row0, hr_row1)."""
        self.ctx.check(load().srmap_problem_set_cost_rows(self._h, hr_row0, hr_row1))

    def set_observations(self, lr):
        a, pa = _d(lr)
        assert a.size == self.K * self.C * self.h * self.w, (a.shape, self.K, self.C, self.h, self.w)
        self.ctx.check(load().srmap_set_observations(self._h, pa))

    def active_impl(self):
        v = C.c_int(-1)
        self.ctx.check(load().srmap_problem_active_impl(self._h, C.byref(v)))
        return v.value

    def set_observations_device(self, ptr, stream=None):
        self.ctx.check(load().srmap_set_observations_device(self._h, C.c_void_p(ptr), C.c_void_p(stream or 0)))

    def add_regularizer(self, kind, lam, btv_range=0, btv_decay=0.0):
        idx = C.c_int(-1)
        self.ctx.check(load().srmap_add_regularizer(self._h, kind, lam, btv_range, btv_decay, C.byref(idx)))
        self.nreg += 1
        return idx.value

    def clear_regularizers(self):
        self.ctx.check(load().srmap_clear_regularizers(self._h))
        self.nreg = 0

    def set_irls_weights(self, reg, w):
        if w is None:
            self.ctx.check(load().srmap_set_irls_weights(self._h, reg, None))
        else:
            a, pa = _d(w)
            assert a.size == self.C * self.H * self.W
            self.ctx.check(load().srmap_set_irls_weights(self._h, reg, pa))

    def update_irls_weights_device(self, reg, x_ptr, stream=None):
        self.ctx.check(load().srmap_update_irls_weights_device(self._h, reg, C.c_void_p(x_ptr), C.c_void_p(stream or 0)))

    def apply(self, hr, k):
        a, pa = _d(hr)
        assert a.size == self.C * self.H * self.W
        out = np.empty((self.C, self.h, self.w))
        self.ctx.check(load().srmap_apply(self._h, k, pa, out.ctypes.data_as(c_double_p)))
        return out

    def apply_transpose(self, lr, k):
        a, pa = _d(lr)
        assert a.size == self.C * self.h * self.w
        out = np.empty((self.C, self.H, self.W))
        self.ctx.check(load().srmap_apply_transpose(self._h, k, pa, out.ctypes.data_as(c_double_p)))
        return out

    def reg_values(self, reg, x):
        a, pa = _d(x)
        out = np.empty((self.C, self.H, self.W))
        self.ctx.check(load().srmap_reg_values(self._h, reg, pa, out.ctypes.data_as(c_double_p)))
        return out

    def reg_values_and_gradient(self, reg, x, gc):
        a, pa = _d(x)
        g, pg = _d(gc)
        vals = np.empty((self.C, self.H, self.W))
        grad = np.empty((self.C, self.H, self.W))
        self.ctx.check(load().srmap_reg_values_and_gradient(
            self._h, reg, pa, pg, vals.ctypes.data_as(c_double_p), grad.ctypes.data_as(c_double_p)))
        return vals, grad

    def eval(self, x, terms=TERM_ALL, want_grad=True):
        a, pa = _d(x)
        assert a.size == self.C * self.H * self.W
        cost = C.c_double()
        g = np.empty((self.C, self.H, self.W)) if want_grad else None
        self.ctx.check(load().srmap_eval(self._h, terms, pa, C.byref(cost),
                                         g.ctypes.data_as(c_double_p) if want_grad else None))
        return cost.value, g

    def eval_device(self, x_ptr, g_ptr, terms=TERM_ALL, want_cost=False, stream=None):
        cost = C.c_double()
        self.ctx.check(load().srmap_eval_device(self._h, terms, C.c_void_p(x_ptr),
                                                C.c_void_p(g_ptr) if g_ptr else None,
                                                C.byref(cost) if want_cost else None,
                                                C.c_void_p(stream) if stream else None))
        return cost.value if want_cost else None

    def last_cost(self):
        cost = C.c_double()
        self.ctx.check(load().srmap_last_cost(self._h, C.byref(cost)))
        return cost.value

    def solve(self, x0, options=None, comm=None, shard=None):
        """IRLSMapSolver::Solve; with a Comm and a ShardDesc: this rank's shard of the joint solve."""
        a, pa = _d(x0)
        assert a.size == self.C * self.H * self.W
        out = np.empty((self.C, self.H, self.W))
        rep = SolveReport()
        o = options if options is not None else default_irls_options()
        if comm is None:
            st = load().srmap_solve(self._h, C.byref(o), pa, out.ctypes.data_as(c_double_p), C.byref(rep))
        else:
            st = load().srmap_solve_sharded(self._h, comm._h, C.byref(shard), C.byref(o), pa,
                                            out.ctypes.data_as(c_double_p), C.byref(rep))
        self.ctx.check(st)
        return out, rep

    def selfcheck(self):
        """Largest relative deviation of the solver's derived beta denominator from the directly summed y.dk (host-paced solves)."""
        v = C.c_double(0.0)
        self.ctx.check(load().srmap_problem_selfcheck(self._h, C.byref(v)))
        return v.value

    def eval_sharded_device(self, comm, shard, x_ptr, g_ptr, terms=TERM_ALL, want_cost=False, stream=None):
        cost = C.c_double()
        self.ctx.check(load().srmap_eval_sharded_device(
            self._h, comm._h if comm is not None else None, C.byref(shard) if shard is not None else None, terms,
            C.c_void_p(x_ptr), C.c_void_p(g_ptr) if g_ptr else None, C.byref(cost) if want_cost else None,
            C.c_void_p(stream) if stream else None))
        return cost.value if want_cost else None

    def cg_trace(self, x0, epsg=0.0, epsf=0.0, epsx=0.0, maxits=0, cap=4096):
        """One nonlinear-CG run; returns (x, iterations, nfev, termination, [f of every evaluation])."""
        a, pa = _d(x0)
        out = np.empty((self.C, self.H, self.W))
        its, nfev, term, tl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        tr = np.zeros(cap)
        self.ctx.check(load().srmap_cg_trace(self._h, epsg, epsf, epsx, maxits, pa, out.ctypes.data_as(c_double_p),
                                             C.byref(its), C.byref(nfev), C.byref(term),
                                             tr.ctypes.data_as(c_double_p), cap, C.byref(tl)))
        return out, its.value, nfev.value, term.value, tr[:min(cap, tl.value)].copy()


class Comm:
    """srmap_comm: RCCL (ranks on different GPUs) or host callbacks over a torch.distributed group (gloo / any)."""

    def __init__(self, ctx, rank, world, backend="rccl", unique_id=None, dist=None, group=None, group_ranks=None):
        self.ctx, self.rank, self.world = ctx, rank, world
        self._h = C.c_void_p()
        self._keep = None
        if backend == "rccl":
            assert unique_id is not None and len(unique_id) == 128
            ctx.check(load().srmap_comm_create_rccl(ctx._h, unique_id, rank, world, C.byref(self._h)))
        else:
            import torch

            def _arr(ptr, count, dtype):
                ct = C.c_float if dtype == F32 else C.c_double
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(count,))

            def _ar(buf, count, dtype, op, _user):
                try:
                    t = torch.from_numpy(_arr(buf, count, dtype))
                    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM, group=group)
                    return 0
                except Exception as e:  # pragma: no cover
                    print("host all-reduce failed:", e)
                    return 1

            def _sr(send, sbytes, dst, recv, rbytes, src, _user):
                try:
                    reqs = []
                    if dst >= 0 and sbytes:
                        ts = torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(sbytes,)).copy())
                        reqs.append(dist.isend(ts, group_ranks[dst] if group_ranks else dst, group=group))
                    tr = None
                    if src >= 0 and rbytes:
                        tr = torch.empty(rbytes, dtype=torch.uint8)
                        reqs.append(dist.irecv(tr, group_ranks[src] if group_ranks else src, group=group))
                    for r in reqs:
                        r.wait()
                    if tr is not None:
                        np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(rbytes,))[:] = tr.numpy()
                    return 0
                except Exception as e:  # pragma: no cover
                    print("host send/recv failed:", e)
                    return 1

            self._keep = (HOST_ALLREDUCE_FN(_ar), HOST_SENDRECV_FN(_sr))
            ctx.check(load().srmap_comm_create_host(ctx._h, rank, world, self._keep[0], self._keep[1], None, C.byref(self._h)))

    def allreduce(self, dev_ptr, count, dtype=F64, op=0, stream=None):
        self.ctx.check(load().srmap_comm_allreduce(self._h, C.c_void_p(dev_ptr), count, dtype, op,
                                                   C.c_void_p(stream) if stream else None))

    def info(self):
        """(rank, size, backend) as the communicator itself reports them (size = ncclCommCount for RCCL; backend 1 = RCCL)."""
        r, w, b = C.c_int(), C.c_int(), C.c_int()
        self.ctx.check(load().srmap_comm_info(self._h, C.byref(r), C.byref(w), C.byref(b)))
        return r.value, w.value, b.value

    def set_overlap(self, on):
        """Row shards: halo exchange under the interior tile rows (default: host backend on, RCCL off)."""
        self.ctx.check(load().srmap_comm_set_overlap(self._h, 1 if on else 0))

    def describe(self):
        """'rccl <version> <path of the loaded librccl>' or 'host callbacks'."""
        buf = C.create_string_buffer(512)
        self.ctx.check(load().srmap_comm_describe(self._h, buf, C.c_size_t(512)))
        return buf.value.decode()

    def split(self, color, key, new_rank, new_world):
        """ncclCommSplit (RCCL backend): the ranks passing the same color form a new communicator."""
        c = Comm.__new__(Comm)
        c.ctx, c.rank, c.world, c._keep = self.ctx, new_rank, new_world, None
        c._h = C.c_void_p()
        self.ctx.check(load().srmap_comm_split(self._h, color, key, new_rank, new_world, C.byref(c._h)))
        return c

    @staticmethod
    def unique_id(ctx):
        buf = C.create_string_buffer(128)
        ctx.check(load().srmap_comm_get_unique_id(ctx._h, buf))
        return buf.raw

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:  # at interpreter exit the module globals may be gone
            _lib.srmap_comm_destroy(self._h)
            self._h = None
