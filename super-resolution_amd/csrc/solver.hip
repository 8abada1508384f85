// solver.hip -- IRLSMapSolver::Solve on the GPU (irls_map_solver.cpp:45-157,
// 192-265) with the nonlinear CG the reference obtains from ALGLIB 3.10.0
// (mincg, default settings: DY/HS hybrid beta, More'-Thuente line search,
// libs/alglib/src/optimization.cpp:17137-17880, alglibinternal.cpp:12313-12632).
//
// Every n-vector (iterate, gradient, directions, line-search base point) lives
// in HBM and is touched only by the kernels below; per evaluation only the
// scalars f and g.d cross PCIe.  The control flow (step selection, stopping
// rules) runs on the host, in double, in ALGLIB's order of operations so that
// the trajectory follows the reference's up to reduction order.
#include <chrono>
#include <utility>
#include <cmath>
#include <cstring>
#include <vector>

#include "srmap_internal.hpp"

namespace srmap {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}

constexpr int kRedBlocks = 1024;

// Up to three dot products in one pass (pairs (a0,b0), (a1,b1), (a2,b2));
// block partials -> part[3][gridDim.x].  Products and sums in double.
template <typename T>
__global__ __launch_bounds__(256) void k_dots(const T* __restrict__ a0, const T* __restrict__ b0,
                                             const T* __restrict__ a1, const T* __restrict__ b1,
                                             const T* __restrict__ a2, const T* __restrict__ b2,
                                             size_t n, double* __restrict__ part) {
  __shared__ double red[3][4];
  double s0 = 0, s1 = 0, s2 = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    s0 += (double)a0[i] * (double)b0[i];
    if (a1) s1 += (double)a1[i] * (double)b1[i];
    if (a2) s2 += (double)a2[i] * (double)b2[i];
  }
  s0 = wsum(s0); s1 = wsum(s1); s2 = wsum(s2);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { red[0][wid] = s0; red[1][wid] = s1; red[2][wid] = s2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const double* r = red[threadIdx.x];
    part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = (r[0] + r[1]) + (r[2] + r[3]);
  }
}

// max |a_i| block partials -> part[gridDim.x]
template <typename T>
__global__ __launch_bounds__(256) void k_absmax(const T* __restrict__ a, size_t n,
                                               double* __restrict__ part) {
  __shared__ double red[4];
  double m = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    m = fmax(m, fabs((double)a[i]));
  m = wmax(m);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) red[wid] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// Second stage: rows x nb partials -> out[rows] (sum or max), fixed order.
// `out` may be host-mapped pinned memory (the CG loop reads it after one stream
// sync, no copy kernels); extra_src, when given, is one more device scalar (the
// cost of the evaluation) forwarded to out[rows].
__global__ __launch_bounds__(256) void k_finish(const double* __restrict__ part, int nb, int rows,
                                               int is_max, double* __restrict__ out,
                                               const double* __restrict__ extra_src) {
  __shared__ double red[4];
  if (extra_src != nullptr && threadIdx.x == 0) out[rows] = extra_src[0];
  for (int r = 0; r < rows; ++r) {
    double v = 0;
    for (int i = threadIdx.x; i < nb; i += 256)
      v = is_max ? fmax(v, part[(size_t)r * nb + i]) : v + part[(size_t)r * nb + i];
    v = is_max ? wmax(v) : wsum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    if (threadIdx.x == 0)
      out[r] = is_max ? fmax(fmax(red[0], red[1]), fmax(red[2], red[3]))
                      : (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// dst = alpha * src
template <typename T>
__global__ void k_scale_copy(T* __restrict__ dst, const T* __restrict__ src, T alpha, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = alpha * src[i];
}
// dst = a + alpha * b
template <typename T>
__global__ void k_axpy_out(T* __restrict__ dst, const T* __restrict__ a, const T* __restrict__ b,
                           T alpha, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = a[i] + alpha * b[i];
}
// dn = -g + beta * dk
template <typename T>
__global__ void k_new_direction(T* __restrict__ dn, const T* __restrict__ g, const T* __restrict__ dk,
                                T beta, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { T v = -g[i]; v += beta * dk[i]; dn[i] = v; }
}
// d = (dk * s1) * s2   (linminnormalized's two scalings, alglibinternal.cpp:12165-12196)
template <typename T>
__global__ void k_scale2(T* __restrict__ d, const T* __restrict__ dk, T s1, T s2, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = (dk[i] * s1) * s2;
}
template <typename T>
__global__ void k_fill(T* __restrict__ d, T v, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = v;
}

template <typename T>
struct DeviceCG {
  srmap_problem* p;
  hipStream_t st;
  size_t n;
  // x, g: current point and gradient; xk/dk: accepted point and direction;
  // d: normalised direction; yk = -g_k (then g_{k+1}-g_k).  The line-search base is xk itself.
  T *x = nullptr, *g = nullptr, *xk = nullptr, *dk = nullptr, *dn = nullptr, *d = nullptr,
    *yk = nullptr;
  double* part = nullptr;  // [3][kRedBlocks]
  double* scal = nullptr;  // [4] device
  double* hs = nullptr;    // host-mapped pinned scalars (ctx->h_scal): written by k_finish, read after a stream sync
  srmap_allreduce_fn ar = nullptr;
  void* user = nullptr;
  int evaluations = 0;

  unsigned blocks() const { return (unsigned)((n + 255) / 256); }
  int nb() const { size_t b = (n + 255) / 256; return (int)(b < (size_t)kRedBlocks ? b : kRedBlocks); }

  int alloc() {
    T** v[] = {&x, &g, &xk, &dk, &dn, &d, &yk};
    for (T** q : v) SRMAP_HIP(p->ctx, hipMalloc((void**)q, n * sizeof(T)));
    SRMAP_HIP(p->ctx, hipMalloc((void**)&part, sizeof(double) * 3 * kRedBlocks));
    SRMAP_HIP(p->ctx, hipMalloc((void**)&scal, sizeof(double) * 4));
    int rc = ensure_staging(p->ctx);
    if (rc) return rc;
    hs = p->ctx->h_scal;
    return SRMAP_OK;
  }
  void release() {
    T* v[] = {x, g, xk, dk, dn, d, yk};
    for (T* q : v) if (q) (void)hipFree(q);
    if (part) (void)hipFree(part);
    if (scal) (void)hipFree(scal);
  }
  int copy(T* dst, const T* src) {
    SRMAP_HIP(p->ctx, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToDevice, st));
    return SRMAP_OK;
  }
  // out[0..count) = dot products of the given pairs (global over ranks)
  int dots(const T* a0, const T* b0, const T* a1, const T* b1, const T* a2, const T* b2, int count,
           double* out) {
    hipLaunchKernelGGL(k_dots<T>, dim3(nb()), dim3(256), 0, st, a0, b0, a1, b1, a2, b2, n, part);
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, st, part, nb(), 3, 0, hs, (const double*)nullptr);
    SRMAP_HIP(p->ctx, hipStreamSynchronize(st));
    double h[3] = {hs[0], hs[1], hs[2]};
    if (ar) ar(h, count, user);
    for (int i = 0; i < count; ++i) out[i] = h[i];
    return SRMAP_OK;
  }
  int absmax(const T* a, double* out) {
    hipLaunchKernelGGL(k_absmax<T>, dim3(nb()), dim3(256), 0, st, a, n, part);
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, st, part, nb(), 1, 1, hs, (const double*)nullptr);
    SRMAP_HIP(p->ctx, hipStreamSynchronize(st));
    *out = hs[0];
    return SRMAP_OK;
  }
  // f, g <- objective at x; dg <- g.d (when d_vec != nullptr)
  int evaluate(double* f, const T* d_vec, double* dg) {
    int rc = srmap_eval_device(p, SRMAP_TERM_ALL, x, g, nullptr, st);
    if (rc) return rc;
    evaluations++;
    double h[2] = {0, 0};
    if (d_vec) {
      hipLaunchKernelGGL(k_dots<T>, dim3(nb()), dim3(256), 0, st, (const T*)g, d_vec, (const T*)nullptr,
                         (const T*)nullptr, (const T*)nullptr, (const T*)nullptr, n, part);
      hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, st, part, nb(), 1, 0, hs, (const double*)p->d_cost);
      SRMAP_HIP(p->ctx, hipStreamSynchronize(st));
      h[1] = hs[0]; h[0] = hs[1];
    } else {
      hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, st, part, 0, 0, 0, hs, (const double*)p->d_cost);
      SRMAP_HIP(p->ctx, hipStreamSynchronize(st));
      h[0] = hs[0];
    }
    if (ar) ar(h, 2, user);
    *f = h[0];
    if (dg) *dg = h[1];
    return SRMAP_OK;
  }
};

static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }

// More'-Thuente safeguarded step (MINPACK-2 dcstep; ALGLIB linmin_mcstep,
// alglibinternal.cpp:12972-13232).
struct Bracket { double stx, fx, dx, sty, fy, dy; };

static double cubic_gamma(double theta, double da, double db, bool clamp0) {
  const double s = dmax(std::fabs(theta), dmax(std::fabs(da), std::fabs(db)));
  double t = (theta / s) * (theta / s) - da / s * (db / s);
  if (clamp0) t = dmax(0.0, t);
  return s * std::sqrt(t);
}

static void mt_step(Bracket* b, double* stp, double fp, double dp, bool* brackt, double stmin,
                    double stmax, int* info) {
  *info = 0;
  if ((*brackt && (*stp <= dmin(b->stx, b->sty) || *stp >= dmax(b->stx, b->sty))) ||
      b->dx * (*stp - b->stx) >= 0 || stmax < stmin)
    return;
  const double sgnd = dp * (b->dx / std::fabs(b->dx));
  bool bound;
  double stpf;
  if (fp > b->fx) {
    *info = 1; bound = true;
    const double theta = 3 * (b->fx - fp) / (*stp - b->stx) + b->dx + dp;
    double gamma = cubic_gamma(theta, b->dx, dp, false);
    if (*stp < b->stx) gamma = -gamma;
    const double pp = gamma - b->dx + theta, q = gamma - b->dx + gamma + dp, r = pp / q;
    const double stpc = b->stx + r * (*stp - b->stx);
    const double stpq = b->stx + b->dx / ((b->fx - fp) / (*stp - b->stx) + b->dx) / 2 * (*stp - b->stx);
    stpf = std::fabs(stpc - b->stx) < std::fabs(stpq - b->stx) ? stpc : stpc + (stpq - stpc) / 2;
    *brackt = true;
  } else if (sgnd < 0) {
    *info = 2; bound = false;
    const double theta = 3 * (b->fx - fp) / (*stp - b->stx) + b->dx + dp;
    double gamma = cubic_gamma(theta, b->dx, dp, false);
    if (*stp > b->stx) gamma = -gamma;
    const double pp = gamma - dp + theta, q = gamma - dp + gamma + b->dx, r = pp / q;
    const double stpc = *stp + r * (b->stx - *stp);
    const double stpq = *stp + dp / (dp - b->dx) * (b->stx - *stp);
    stpf = std::fabs(stpc - *stp) > std::fabs(stpq - *stp) ? stpc : stpq;
    *brackt = true;
  } else if (std::fabs(dp) < std::fabs(b->dx)) {
    *info = 3; bound = true;
    const double theta = 3 * (b->fx - fp) / (*stp - b->stx) + b->dx + dp;
    double gamma = cubic_gamma(theta, b->dx, dp, true);
    if (*stp > b->stx) gamma = -gamma;
    const double pp = gamma - dp + theta, q = gamma + (b->dx - dp) + gamma, r = pp / q;
    double stpc;
    if (r < 0 && gamma != 0) stpc = *stp + r * (b->stx - *stp);
    else stpc = *stp > b->stx ? stmax : stmin;
    const double stpq = *stp + dp / (dp - b->dx) * (b->stx - *stp);
    if (*brackt) stpf = std::fabs(*stp - stpc) < std::fabs(*stp - stpq) ? stpc : stpq;
    else stpf = std::fabs(*stp - stpc) > std::fabs(*stp - stpq) ? stpc : stpq;
  } else {
    *info = 4; bound = false;
    if (*brackt) {
      const double theta = 3 * (fp - b->fy) / (b->sty - *stp) + b->dy + dp;
      double gamma = cubic_gamma(theta, b->dy, dp, false);
      if (*stp > b->sty) gamma = -gamma;
      const double pp = gamma - dp + theta, q = gamma - dp + gamma + b->dy, r = pp / q;
      stpf = *stp + r * (b->sty - *stp);
    } else {
      stpf = *stp > b->stx ? stmax : stmin;
    }
  }
  if (fp > b->fx) {
    b->sty = *stp; b->fy = fp; b->dy = dp;
  } else {
    if (sgnd < 0.0) { b->sty = b->stx; b->fy = b->fx; b->dy = b->dx; }
    b->stx = *stp; b->fx = fp; b->dx = dp;
  }
  stpf = dmin(stmax, stpf);
  stpf = dmax(stmin, stpf);
  *stp = stpf;
  if (*brackt && bound) {
    if (b->sty > b->stx) *stp = dmin(b->stx + 0.66 * (b->sty - b->stx), *stp);
    else *stp = dmax(b->stx + 0.66 * (b->sty - b->stx), *stp);
  }
}

// mcsrch with the device evaluation inlined (constants alglibinternal.cpp:156-160;
// trimfunction after each evaluation as mincgiteration does, optimization.cpp:17594).
template <typename T>
static int line_search(DeviceCG<T>& cg, double* f, double dginit, double* stp, double gtol,
                       int* info, int* nfev, double trim) {
  const double ftol = 0.001, xtol = 100 * 5E-16, stpmin = 1.0e-50, stpmax = 1.0e+50, p5 = 0.5,
               p66 = 0.66, xtrapf = 4.0;
  const int maxfev = 20;
  if (*stp < stpmin) *stp = stpmin;
  if (*stp > stpmax) *stp = stpmax;
  int infoc = 1;
  *info = 0;
  *nfev = 0;
  // On entry the base point is cg.xk; cg.x is scratch for the trial points.  The paths that try nothing
  // still leave x = base, as mcsrch does.
  if (*stp <= 0) return cg.copy(cg.x, cg.xk);
  if (dginit >= 0) return cg.copy(cg.x, cg.xk);  // not a descent direction
  bool brackt = false, stage1 = true;
  const double finit = *f, dgtest = ftol * dginit;
  double width = stpmax - stpmin, width1 = width / p5;
  int rc = SRMAP_OK;
  Bracket b = {0, finit, dginit, 0, finit, dginit};
  double stmin = 0, stmax = 0;
  for (;;) {
    if (brackt) { stmin = dmin(b.stx, b.sty); stmax = dmax(b.stx, b.sty); }
    else { stmin = b.stx; stmax = *stp + xtrapf * (*stp - b.stx); }
    if (*stp > stpmax) *stp = stpmax;
    if (*stp < stpmin) *stp = stpmin;
    if ((brackt && (*stp <= stmin || *stp >= stmax)) || *nfev >= maxfev - 1 || infoc == 0 ||
        (brackt && stmax - stmin <= xtol * stmax))
      *stp = b.stx;
    hipLaunchKernelGGL(k_axpy_out<T>, dim3(cg.blocks()), dim3(256), 0, cg.st, cg.x, (const T*)cg.xk,
                       (const T*)cg.d, (T)*stp, cg.n);
    double dg = 0;
    rc = cg.evaluate(f, cg.d, &dg);
    if (rc) return rc;
    if (*f >= trim) {  // trimfunction: F = threshold, G = 0
      *f = trim;
      hipLaunchKernelGGL(k_fill<T>, dim3(cg.blocks()), dim3(256), 0, cg.st, cg.g, T(0), cg.n);
      dg = 0;
    }
    *info = 0;
    *nfev += 1;
    const double ftest1 = finit + *stp * dgtest;
    if ((brackt && (*stp <= stmin || *stp >= stmax)) || infoc == 0) *info = 6;
    if (*stp == stpmax && *f < finit && *f <= ftest1 && dg <= dgtest) *info = 5;
    if (*stp == stpmin && (*f >= finit || *f > ftest1 || dg >= dgtest)) *info = 4;
    if (*nfev >= maxfev) *info = 3;
    if (brackt && stmax - stmin <= xtol * stmax) *info = 2;
    if (*f < finit && *f <= ftest1 && std::fabs(dg) <= -gtol * dginit) *info = 1;
    if (*info != 0) {
      if (*info == 1 || *info == 5) {
        // ALGLIB additionally demotes to 6 when the point did not move
        // (sum (wa-x)^2 == 0); with stp > 0 and a unit d this cannot be 0
        // unless stp*d underflows against x, which we test through stp.
        if (*f >= finit || *stp == 0.0) *info = 6;
      }
      return SRMAP_OK;
    }
    if (stage1 && *f <= ftest1 && dg >= dmin(ftol, gtol) * dginit) stage1 = false;
    if (stage1 && *f <= b.fx && *f > ftest1) {
      const double fm = *f - *stp * dgtest;
      Bracket m = {b.stx, b.fx - b.stx * dgtest, b.dx - dgtest, b.sty, b.fy - b.sty * dgtest, b.dy - dgtest};
      mt_step(&m, stp, fm, dg - dgtest, &brackt, stmin, stmax, &infoc);
      b.stx = m.stx; b.sty = m.sty;
      b.fx = m.fx + m.stx * dgtest; b.fy = m.fy + m.sty * dgtest;
      b.dx = m.dx + dgtest; b.dy = m.dy + dgtest;
    } else {
      mt_step(&b, stp, *f, dg, &brackt, stmin, stmax, &infoc);
    }
    if (brackt) {
      if (std::fabs(b.sty - b.stx) >= p66 * width1) *stp = b.stx + p5 * (b.sty - b.stx);
      width1 = width;
      width = std::fabs(b.sty - b.stx);
    }
  }
}

struct CgResult { int type = 0, its = 0, nfev = 0; double f = 0; };

// mincgiteration (optimization.cpp:17137-17880), default configuration: no
// preconditioner, unit scales, cgtype = 1, no stpmax.  On return cg.x holds
// the accepted point XN.
template <typename T>
static int run_cg(DeviceCG<T>& cg, double epsg, double epsf, double epsx, int maxits, CgResult* out) {
  const double gtol = 0.3;
  const int rscountdownlen = 10;
  if (epsg == 0 && epsf == 0 && epsx == 0 && maxits == 0) epsx = 1.0E-6;
  const size_t n = cg.n;
  const unsigned nbk = cg.blocks();
  hipStream_t st = cg.st;
  CgResult res;
  double f = 0, gg = 0;
  int rc = cg.copy(cg.xk, cg.x);
  if (rc) return rc;
  rc = cg.evaluate(&f, nullptr, nullptr);
  if (rc) return rc;
  const double trim = 10 * (std::fabs(f) + 1);
  hipLaunchKernelGGL(k_scale_copy<T>, dim3(nbk), dim3(256), 0, st, cg.dk, (const T*)cg.g, T(-1), n);
  rc = cg.dots(cg.g, cg.g, nullptr, nullptr, nullptr, nullptr, 1, &gg);
  if (rc) return rc;
  if (std::sqrt(gg) <= epsg) { res.type = 4; res.f = f; *out = res; return cg.copy(cg.x, cg.xk); }
  res.nfev = 1;
  double fold = f, lastgoodstep = 1.0;
  int rstimer = rscountdownlen;
  for (;;) {
    // yk = -g ; d = normalised dk ; x = xk
    hipLaunchKernelGGL(k_scale_copy<T>, dim3(nbk), dim3(256), 0, st, cg.yk, (const T*)cg.g, T(-1), n);
    // (x = xk is not materialised: the line search writes every trial point x = xk + stp * d itself)
    double stp = 1.0, dginit = 0;
    {
      // linminnormalized: d *= 1/max|d| ; d *= 1/sqrt(d.d).  Under a multi-rank
      // allreduce hook (sum only) the first scaling uses the 2-norm instead of
      // the max norm; both only guard against overflow of the squares.
      double mx = 0;
      if (cg.ar) {
        double s2 = 0;
        rc = cg.dots(cg.dk, cg.dk, nullptr, nullptr, nullptr, nullptr, 1, &s2);
        if (rc) return rc;
        mx = std::sqrt(s2);
      } else {
        rc = cg.absmax(cg.dk, &mx);
        if (rc) return rc;
      }
      if (mx != 0) {
        const double s1 = 1 / mx;
        stp = stp / s1;
        // sum (dk*s1)^2 evaluated on the scaled vector
        hipLaunchKernelGGL(k_scale_copy<T>, dim3(nbk), dim3(256), 0, st, cg.d, (const T*)cg.dk, (T)s1, n);
        double ss = 0;
        rc = cg.dots(cg.d, cg.d, nullptr, nullptr, nullptr, nullptr, 1, &ss);
        if (rc) return rc;
        const double s2 = 1 / std::sqrt(ss);
        hipLaunchKernelGGL(k_scale2<T>, dim3(nbk), dim3(256), 0, st, cg.d, (const T*)cg.dk, (T)s1, (T)s2, n);
        stp = stp / s2;
      } else {
        rc = cg.copy(cg.d, cg.dk);
        if (rc) return rc;
      }
    }
    if (lastgoodstep != 0) stp = lastgoodstep;
    double dd = 0;
    {
      double h[2];
      rc = cg.dots(cg.g, cg.d, cg.d, cg.d, nullptr, nullptr, 2, h);
      if (rc) return rc;
      dginit = h[0];
      dd = h[1];
    }
    int mcinfo = 0, nfev = 0;
    rc = line_search(cg, &f, dginit, &stp, gtol, &mcinfo, &nfev, trim);
    if (rc) return rc;
    double betak = 0;
    double dots3[3] = {0, 0, 0};
    if (mcinfo == 1) {
      // yk += g ; vv = yk.dk ; betady = g.g/vv ; betahs = g.yk/vv
      hipLaunchKernelGGL(k_axpy_out<T>, dim3(nbk), dim3(256), 0, st, cg.yk, (const T*)cg.yk, (const T*)cg.g, T(1), n);
      rc = cg.dots(cg.yk, cg.dk, cg.g, cg.g, cg.g, cg.yk, 3, dots3);
      if (rc) return rc;
      const double vv = dots3[0];
      betak = dmax(0.0, dmin(dots3[1] / vv, dots3[2] / vv));
      gg = dots3[1];
    } else {
      rc = cg.dots(cg.g, cg.g, nullptr, nullptr, nullptr, nullptr, 1, &gg);
      if (rc) return rc;
    }
    if (res.its > 0 && res.its % (3 + (long long)n) == 0) betak = 0;
    if (mcinfo == 1 || mcinfo == 5) rstimer = rscountdownlen; else rstimer -= 1;
    hipLaunchKernelGGL(k_new_direction<T>, dim3(nbk), dim3(256), 0, st, cg.dn, (const T*)cg.g, (const T*)cg.dk, (T)betak, n);
    const double lastscaledstep = stp * std::sqrt(dd);
    if (mcinfo == 1) lastgoodstep = stp * std::sqrt(dd);
    if (!std::isfinite(gg) || !std::isfinite(f)) { res.type = -8; break; }
    res.nfev += nfev;
    res.its += 1;
    if (res.its >= maxits && maxits > 0) { res.type = 5; break; }
    if (std::sqrt(gg) <= epsg) { res.type = 4; break; }
    if (fold - f <= epsf * dmax(std::fabs(fold), dmax(std::fabs(f), 1.0))) { res.type = 1; break; }
    if (lastscaledstep <= epsx) { res.type = 2; break; }
    if (rstimer <= 0) { res.type = 7; break; }
    std::swap(cg.xk, cg.x);    // xk <- accepted point; the old xk becomes trial scratch
    std::swap(cg.dk, cg.dn);   // dk <- new direction
    fold = f;
  }
  res.f = f;
  *out = res;
  return SRMAP_OK;
}

template <typename T>
static int solve_typed(srmap_problem* p, const srmap_irls_options* opt, const double* x0, double* x_out,
                       srmap_solve_report* report, srmap_allreduce_fn ar, void* user) {
  if (!p->have_obs) return set_error(p->ctx, SRMAP_EINVAL, "cannot super-resolve with 0 low-res images");
  const Geometry& geo = p->geo;
  const size_t N = (size_t)geo.W * geo.H;
  const int C = geo.C;
  const int per_split = opt->split_channels ? 1 : C;
  const int rounds = C / per_split;
  const size_t npts = (size_t)per_split * N;
  srmap_irls_options o = *opt;
  double lambda_sum = 0.0;
  for (int r = 0; r < p->nreg; ++r) lambda_sum += p->reg[r].lambda;
  {  // AdjustThresholdsAdaptively (map_solver.cpp:16-26, irls_map_solver.cpp:161-171)
    const double scale = (double)(int)npts * lambda_sum;
    if (!(scale < 1.0)) {
      o.gradient_norm_threshold *= scale;
      o.cost_decrease_threshold *= scale;
      o.parameter_variation_threshold *= scale;
      o.irls_cost_difference_threshold *= scale;
    }
  }
  srmap_solve_report rep = {0, 0, 0, 0, 0.0};
  hipStream_t st = p->ctx->stream;
  static const bool timing = getenv("SRMAP_DEBUG_SOLVE_TIMING") != nullptr;  // phase wall times (profiling aid)
  auto now = [&]() { if (timing) (void)hipStreamSynchronize(st); return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count(); };
  double t_alloc = 0, t_up = 0, t_loop = 0, t_down = 0;
  auto t0 = now();
  DeviceCG<T> cg;
  cg.p = p; cg.st = st; cg.n = npts; cg.ar = ar; cg.user = user;
  int rc = cg.alloc();
  // IRLS weights live in the problem's RegSpec (full [C][H][W]); make sure they exist.
  for (int r = 0; r < p->nreg && rc == SRMAP_OK; ++r) {
    if (!p->reg[r].weights) {
      hipError_t e = hipMalloc(&p->reg[r].weights, p->hr_count() * sizeof(T));
      if (e != hipSuccess) rc = set_error(p->ctx, SRMAP_ENOMEM, "hipMalloc failed");
    }
  }
  T* regvals = nullptr;
  if (rc == SRMAP_OK && p->nreg > 0 && hipMalloc((void**)&regvals, npts * sizeof(T)) != hipSuccess)
    rc = set_error(p->ctx, SRMAP_ENOMEM, "hipMalloc failed");
  const int saved_c0 = p->view_c0, saved_C = p->view_C;
  t_alloc = ms(t0, now());
  for (int round = 0; round < rounds && rc == SRMAP_OK; ++round) {
    const int c0 = round * per_split;
    p->view_c0 = c0;
    p->view_C = per_split;
    Geometry vg = geo;
    vg.C = per_split;
    auto t1 = now();
    rc = convert_upload(p, x0 + (size_t)c0 * N, cg.x, npts, st);
    if (rc) break;
    auto t2 = now();
    t_up += ms(t1, t2);
    // w <- 1  (irls_map_solver.cpp:66-74)
    for (int r = 0; r < p->nreg; ++r)
      hipLaunchKernelGGL(k_fill<T>, dim3(cg.blocks()), dim3(256), 0, st, (T*)p->reg[r].weights + (size_t)c0 * N, T(1), npts);
    double previous_cost = INFINITY;
    double cost_difference = o.irls_cost_difference_threshold + 1.0;
    int ran = 0;
    while (std::fabs(cost_difference) >= o.irls_cost_difference_threshold) {
      CgResult cr;
      rc = run_cg(cg, o.gradient_norm_threshold, o.cost_decrease_threshold, o.parameter_variation_threshold,
                  o.max_num_solver_iterations, &cr);
      if (rc) break;
      rep.cg_iterations += cr.its;
      rep.last_termination = cr.type;
      rep.final_cost = cr.f;
      if (p->nreg == 0) { ran++; break; }
      for (int r = 0; r < p->nreg; ++r) {  // w = 1/max(1e-5, reg(x)), :128-143
        rc = launch_reg_values<T>(p, vg, p->reg[r], (const T*)cg.x, regvals, st);
        if (rc) break;
        rc = launch_irls_weights<T>(p, (const T*)regvals, (T*)p->reg[r].weights + (size_t)c0 * N, npts, st);
        if (rc) break;
      }
      if (rc) break;
      cost_difference = previous_cost - cr.f;
      previous_cost = cr.f;
      ran++;
      if (o.max_num_irls_iterations > 0 && ran >= o.max_num_irls_iterations) break;
    }
    if (rc) break;
    rep.irls_rounds += ran;
    auto t3 = now();
    t_loop += ms(t2, t3);
    rc = convert_download(p, cg.x, x_out + (size_t)c0 * N, npts, st);
    t_down += ms(t3, now());
  }
  rep.evaluations = cg.evaluations;
  p->view_c0 = saved_c0;
  p->view_C = saved_C;
  auto t4 = now();
  if (regvals) (void)hipFree(regvals);
  cg.release();
  if (timing)
    fprintf(stderr, "[solve] alloc %.2f ms, upload %.2f, irls/cg loop %.2f (%d evaluations), download %.2f, free %.2f\n", t_alloc,
            t_up, t_loop, cg.evaluations, t_down, ms(t4, now()));
  if (report) *report = rep;
  return rc;
}

int solve_impl(srmap_problem* p, const srmap_irls_options* o, const double* x0, double* x_out,
               srmap_solve_report* rep, srmap_allreduce_fn ar, void* user) {
  if (p->dtype == SRMAP_F32) return solve_typed<float>(p, o, x0, x_out, rep, ar, user);
  return solve_typed<double>(p, o, x0, x_out, rep, ar, user);
}

}  // namespace srmap
