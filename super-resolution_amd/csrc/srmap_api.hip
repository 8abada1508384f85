// srmap_api.hip -- C ABI (include/srmap.h) of the MI355X MAP gradient path:
// problem set-up (what the reference's ImageModel / MapSolver constructors do
// on the host), buffer management, and the dispatch of one cost+gradient
// evaluation onto the HIP kernels.  No CPU compute path exists here: every
// numeric entry point launches gfx950 kernels or fails.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <new>

#include "srmap_internal.hpp"

using namespace srmap;

namespace srmap {

int set_error(srmap_ctx* ctx, int status, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->error = buf;
  return status;
}

// ---- host restatement of the OpenCV 3.x parameter arithmetic the reference
// relies on (the per-pixel work itself happens in the kernels) ----

static inline int cv_round(double v) { return (int)std::lrint(v); }

// cv::warpAffine fixed-point coordinate tables for M = [1 0 dx; 0 1 dy]
// (motion_module.cpp:18-25): AB_BITS = 10, INTER_BITS = 5.
static void warp_tables(int W, int H, double dx, double dy, std::vector<int>* X,
                        std::vector<int>* Y) {
  double M[6] = {1.0, 0.0, dx, 0.0, 1.0, dy};
  double D = M[0] * M[4] - M[1] * M[3];
  D = D != 0 ? 1.0 / D : 0;
  const double A11 = M[4] * D, A22 = M[0] * D;
  M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
  const double b1 = -M[0] * M[2] - M[1] * M[5];
  const double b2 = -M[3] * M[2] - M[4] * M[5];
  M[2] = b1; M[5] = b2;
  X->resize(W);
  Y->resize(H);
  for (int x = 0; x < W; ++x)
    (*X)[x] = (cv_round((M[1] * 0 + M[2]) * 1024) + 16 + cv_round(M[0] * x * 1024)) >> 5;
  for (int y = 0; y < H; ++y)
    (*Y)[y] = (cv_round((M[4] * y + M[5]) * 1024) + 16 + cv_round(M[3] * 0 * 1024)) >> 5;
}

static int make_warp(srmap_ctx* ctx, int W, int H, double dx, double dy,
                     WarpTaps<double>* out, std::vector<int>* ytab) {
  if (!(std::fabs(dx) < 16000.0) || !(std::fabs(dy) < 16000.0))
    return set_error(ctx, SRMAP_EUNSUPPORTED, "motion shift (%g, %g) too large", dx, dy);
  std::vector<int> X, Y;
  warp_tables(W, H, dx, dy, &X, &Y);
  for (int x = 0; x < W; ++x)
    if (X[x] != X[0] + 32 * x)  // adelta[x] + a constant: cannot happen for a pure translation
      return set_error(ctx, SRMAP_EUNSUPPORTED, "non-uniform warpAffine x table");
  bool uniform_y = true;
  for (int y = 0; y < H; ++y)
    if (Y[y] != Y[0] + 32 * y) uniform_y = false;
  out->ox = X[0] >> 5;
  out->oy = Y[0] >> 5;
  const int fx = X[0] & 31, fy = Y[0] & 31;
  // BilinearTab_f: float32 table; the products are exact multiples of 1/1024.
  const float tx1 = (float)fx * (1.f / 32), tx0 = 1.f - tx1;
  const float ty1 = (float)fy * (1.f / 32), ty0 = 1.f - ty1;
  out->w[0] = ty0 * tx0; out->w[1] = ty0 * tx1; out->w[2] = ty1 * tx0; out->w[3] = ty1 * tx1;
  out->ntaps = (fx == 0 && fy == 0) ? 1 : 4;
  out->fx = fx;
  out->ytab = nullptr;
  ytab->clear();
  if (!uniform_y) {
    // dy within floating-point rounding of a 1/32-px quantisation tie: warpAffine's per-row y coordinate
    // (cvRound((y + b) * 1024) evaluated in double) lands on either side of the tie depending on the row.  The
    // kernels then read the source row and fraction of every destination row from a table.
    ytab->resize(H);
    for (int y = 0; y < H; ++y) (*ytab)[y] = Y[y];  // absolute: source row << 5 | fraction
    out->ntaps = 4;
  }
  return SRMAP_OK;
}

// cv::resize(INTER_NEAREST) source index per destination index
// (image_data.cpp:338-350).
static void nearest_map(int src_len, int dst_len, std::vector<int>* map) {
  const double inv_scale = (double)dst_len / src_len;
  const double ifx = 1.0 / inv_scale;
  map->resize(dst_len);
  for (int x = 0; x < dst_len; ++x) {
    const int sx = (int)std::floor(x * ifx);
    (*map)[x] = sx < src_len - 1 ? sx : src_len - 1;
  }
}

template <typename T>
static int upload_warps(srmap_problem* p, const std::vector<WarpTaps<double>>& src, void** dst) {
  std::vector<WarpTaps<T>> tmp(src.size());
  for (size_t i = 0; i < src.size(); ++i) {
    tmp[i].ox = src[i].ox; tmp[i].oy = src[i].oy; tmp[i].ntaps = src[i].ntaps; tmp[i].fx = src[i].fx;
    tmp[i].ytab = src[i].ytab;
    for (int t = 0; t < 4; ++t) tmp[i].w[t] = (T)src[i].w[t];
  }
  SRMAP_HIP(p->ctx, hipMalloc(dst, sizeof(WarpTaps<T>) * tmp.size()));
  SRMAP_HIP(p->ctx, hipMemcpy(*dst, tmp.data(), sizeof(WarpTaps<T>) * tmp.size(), hipMemcpyHostToDevice));
  return SRMAP_OK;
}

template <typename T>
static int upload_array(srmap_problem* p, const std::vector<double>& src, void** dst) {
  std::vector<T> tmp(src.begin(), src.end());
  SRMAP_HIP(p->ctx, hipMalloc(dst, sizeof(T) * tmp.size()));
  SRMAP_HIP(p->ctx, hipMemcpy(*dst, tmp.data(), sizeof(T) * tmp.size(), hipMemcpyHostToDevice));
  return SRMAP_OK;
}

template <typename T>
__global__ void k_from_double(const double* __restrict__ src, T* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (T)src[i];
}
template <typename T>
__global__ void k_to_double(const T* __restrict__ src, double* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (double)src[i];
}

// Caller buffers are pageable.  A plain hipMemcpy of them runs at 2-6 GB/s; two
// pinned chunks pipeline the PCIe transfer against the host-side memcpy.
constexpr size_t kStageBytes = 4u << 20;

int ensure_staging(srmap_ctx* ctx) {
  for (int i = 0; i < 2; ++i) {
    if (!ctx->h_stage[i]) SRMAP_HIP(ctx, hipHostMalloc(&ctx->h_stage[i], kStageBytes, hipHostMallocDefault));
    if (!ctx->h_event[i]) SRMAP_HIP(ctx, hipEventCreateWithFlags(&ctx->h_event[i], hipEventDisableTiming));
  }
  if (!ctx->h_scal) {
    SRMAP_HIP(ctx, hipHostMalloc((void**)&ctx->h_scal, 16 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    for (int i = 0; i < 16; ++i) ctx->h_scal[i] = 0.0;
  }
  return SRMAP_OK;
}

static int staged_h2d(srmap_ctx* ctx, void* dev, const void* host, size_t bytes, hipStream_t st) {
  int rc = ensure_staging(ctx);
  if (rc) return rc;
  size_t off = 0;
  for (int i = 0; off < bytes; ++i, off += kStageBytes) {
    const size_t nb = bytes - off < kStageBytes ? bytes - off : kStageBytes;
    const int b = i & 1;
    if (i >= 2) SRMAP_HIP(ctx, hipEventSynchronize(ctx->h_event[b]));  // chunk i-2 has left the buffer
    std::memcpy(ctx->h_stage[b], (const char*)host + off, nb);
    SRMAP_HIP(ctx, hipMemcpyAsync((char*)dev + off, ctx->h_stage[b], nb, hipMemcpyHostToDevice, st));
    SRMAP_HIP(ctx, hipEventRecord(ctx->h_event[b], st));
  }
  SRMAP_HIP(ctx, hipStreamSynchronize(st));
  return SRMAP_OK;
}

static int staged_d2h(srmap_ctx* ctx, void* host, const void* dev, size_t bytes, hipStream_t st) {
  int rc = ensure_staging(ctx);
  if (rc) return rc;
  const size_t nchunks = (bytes + kStageBytes - 1) / kStageBytes;
  for (size_t i = 0; i <= nchunks; ++i) {
    if (i < nchunks) {  // chunk i -> pinned buffer (its previous content, chunk i-2, was copied out below)
      const size_t off = i * kStageBytes, nb = bytes - off < kStageBytes ? bytes - off : kStageBytes;
      SRMAP_HIP(ctx, hipMemcpyAsync(ctx->h_stage[i & 1], (const char*)dev + off, nb, hipMemcpyDeviceToHost, st));
      SRMAP_HIP(ctx, hipEventRecord(ctx->h_event[i & 1], st));
    }
    if (i >= 1) {  // chunk i-1 -> caller, while chunk i is on the wire
      const size_t off = (i - 1) * kStageBytes, nb = bytes - off < kStageBytes ? bytes - off : kStageBytes;
      SRMAP_HIP(ctx, hipEventSynchronize(ctx->h_event[(i - 1) & 1]));
      std::memcpy((char*)host + off, ctx->h_stage[(i - 1) & 1], nb);
    }
  }
  return SRMAP_OK;
}

int convert_upload(srmap_problem* p, const double* host, void* dev, size_t n, hipStream_t st) {
  if (p->dtype == SRMAP_F64) return staged_h2d(p->ctx, dev, host, n * 8, st);
  double* tmp = nullptr;
  SRMAP_HIP(p->ctx, hipMalloc((void**)&tmp, n * 8));
  int rc = staged_h2d(p->ctx, tmp, host, n * 8, st);
  hipError_t e = hipSuccess;
  if (rc == SRMAP_OK) {
    hipLaunchKernelGGL(k_from_double<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       tmp, (float*)dev, n);
    e = hipStreamSynchronize(st);
  }
  (void)hipFree(tmp);
  if (rc) return rc;
  SRMAP_HIP(p->ctx, e);
  return SRMAP_OK;
}

int convert_download(srmap_problem* p, const void* dev, double* host, size_t n, hipStream_t st) {
  if (p->dtype == SRMAP_F64) return staged_d2h(p->ctx, host, dev, n * 8, st);
  double* tmp = nullptr;
  SRMAP_HIP(p->ctx, hipMalloc((void**)&tmp, n * 8));
  hipLaunchKernelGGL(k_to_double<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                     (const float*)dev, tmp, n);
  int rc = staged_d2h(p->ctx, host, tmp, n * 8, st);
  (void)hipFree(tmp);
  return rc;
}

static int ensure(srmap_problem* p, void** buf, size_t bytes) {
  if (*buf) return SRMAP_OK;
  SRMAP_HIP(p->ctx, hipMalloc(buf, bytes ? bytes : 8));
  return SRMAP_OK;
}

static int ensure_partials(srmap_problem* p, size_t n) {
  n *= 2;  // second half: partials of g.d (eval_dvec)
  if (p->partials_cap >= n) return SRMAP_OK;
  if (p->d_partials) (void)hipFree(p->d_partials);
  p->d_partials = nullptr;
  p->partials_cap = 0;
  SRMAP_HIP(p->ctx, hipMalloc((void**)&p->d_partials, n * sizeof(double)));
  p->partials_cap = n;
  return SRMAP_OK;
}

static size_t partials_needed(const srmap_problem* p) {
  const Geometry& g = p->geo;
  const size_t fwd = (size_t)((g.w * g.h + 255) / 256) * g.C * g.K;
  const size_t reg_blocks = std::max((size_t)((g.W * g.H + 255) / 256), (size_t)((g.W + 63) / 64) * ((g.H + 3) / 4));
  const size_t reg = reg_blocks * g.C * kMaxRegularizers;
  const size_t n = fwd + reg + ztile_partials_needed(p) + 16;
  return n + (size_t)reduce_scratch_slots(n) + 16;  // + the second-stage scratch of launch_reduce_partials
}

// One ObjectiveFunction::ComputeAllTerms on device buffers.
template <typename T>
static int eval_typed(srmap_problem* p, unsigned terms, const T* x, T* g, hipStream_t st) {
  Geometry geo = p->geo;
  const int c0 = p->view_C > 0 ? p->view_c0 : 0;
  if (p->view_C > 0) geo.C = p->view_C;
  geo.zlo = (p->view_coupled && c0 > 0) ? 1 : 0;
  geo.zhi = (p->view_coupled && c0 + geo.C < p->geo.C) ? 1 : 0;
  const size_t N = (size_t)geo.W * geo.H;
  int rc = ensure_partials(p, partials_needed(p));
  if (rc) return rc;
  if ((terms & SRMAP_TERM_DATA) && !p->have_obs)
    return set_error(p->ctx, SRMAP_EINVAL, "data term requested but no observations set");
  const bool ztile = p->impl != SRMAP_IMPL_DIRECT && p->zplan != nullptr;
  if (p->ov_hook != nullptr && !(ztile && ztile_overlaps_halo(p))) {
    // row shard on a path that cannot run under the halo exchange: exchange first
    int (*hook)(void*) = p->ov_hook;
    p->ov_hook = nullptr;
    rc = hook(p->ov_arg);
    if (rc) return rc;
    SRMAP_HIP(p->ctx, hipStreamWaitEvent(st, p->ov_event, 0));
  }
  if (p->impl == SRMAP_IMPL_TILED && !ztile)
    return set_error(p->ctx, SRMAP_EUNSUPPORTED, "tiled kernels do not cover this geometry");
  int nparts = 0;
  if (ztile) {
    rc = launch_eval_ztile<T>(p, geo, c0, terms, x, g, p->d_partials, &nparts, st);
    if (rc) return rc;
  } else {
    bool g_written = false;
    if (terms & SRMAP_TERM_DATA) {
      rc = ensure(p, &p->d_resid, p->lr_count() * sizeof(T));
      if (rc) return rc;
      int nb = 0;
      rc = launch_forward_direct<T>(p, geo, x, (const T*)p->d_obs, p->geo.C, c0,
                                    (T*)p->d_resid, 0, geo.K, p->d_partials + nparts, &nb, st);
      if (rc) return rc;
      nparts += nb;
      if (g) {
        rc = launch_gather_direct<T>(p, geo, (const T*)p->d_resid, g, 0, geo.K,
                                     2.0 * geo.s * geo.s, false, st);
        if (rc) return rc;
        g_written = true;
      }
    }
    if (g && !g_written) SRMAP_HIP(p->ctx, hipMemsetAsync(g, 0, N * geo.C * sizeof(T), st));
    if (terms & SRMAP_TERM_REG) {
      for (int r = 0; r < p->nreg; ++r) {
        const RegSpec& rs = p->reg[r];
        if (rs.lambda <= 0.0) continue;  // objective_irls_regularization_term.cpp:16-18
        if (!g) {
          // cost only: the gradient kernel still produces the lambda*w*r^2 partials
        }
        const bool onfly = rs.kind != SRMAP_REG_BTV;  // TV kinds: values recomputed in the gradient kernel
        if (!onfly) {
          rc = ensure(p, &p->d_regvals, p->hr_count() * sizeof(T));
          if (rc) return rc;
          rc = launch_reg_values<T>(p, geo, rs, x, (T*)p->d_regvals, st);
          if (rc) return rc;
        }
        int nb = 0;
        const T* wts = rs.weights ? (const T*)rs.weights + (size_t)c0 * N : nullptr;
        rc = launch_reg_gradient_direct<T>(p, geo, rs, x, wts, rs.lambda,
                                           onfly ? nullptr : (const T*)p->d_regvals, g, true,
                                           p->d_partials + nparts, &nb, st);
        if (rc) return rc;
        nparts += nb;
      }
    }
  }
  if (ztile && nparts == 0) return SRMAP_OK;  // reduced inside the last kernel of the evaluation
  return launch_reduce_partials(p, p->d_partials, nparts, p->d_cost, st);
}

// ---- stream ordering of the problem's device state (include/srmap.h, "Streams") ----
// before overwriting observations / weights: the last evaluation may still be reading them on another stream
static int state_begin_write(srmap_problem* p, hipStream_t st) {
  if (p->use_stream && p->use_stream != st) SRMAP_HIP(p->ctx, hipStreamSynchronize(p->use_stream));
  return SRMAP_OK;
}
// after an ASYNCHRONOUS write on st: later evaluations on other streams wait for this event
static int state_end_write(srmap_problem* p, hipStream_t st) {
  if (!p->state_ev) SRMAP_HIP(p->ctx, hipEventCreateWithFlags(&p->state_ev, hipEventDisableTiming));
  SRMAP_HIP(p->ctx, hipEventRecord(p->state_ev, st));
  p->state_stream = st;
  p->state_seen = nullptr;
  return SRMAP_OK;
}
// an evaluation on st: ordered after the last asynchronous state write (nothing to do on the same stream)
static inline int state_read(srmap_problem* p, hipStream_t st) {
  if (p->state_stream && p->state_stream != st && p->state_seen != st) {
    SRMAP_HIP(p->ctx, hipStreamWaitEvent(st, p->state_ev, 0));
    p->state_seen = st;
  }
  p->use_stream = st;
  return SRMAP_OK;
}

static int eval_dispatch(srmap_problem* p, unsigned terms, const void* x, void* g,
                         hipStream_t st) {
  if (int rc = state_read(p, st)) return rc;
  if (p->dtype == SRMAP_F32) return eval_typed<float>(p, terms, (const float*)x, (float*)g, st);
  return eval_typed<double>(p, terms, (const double*)x, (double*)g, st);
}

}  // namespace srmap

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

const char* srmap_version(void) { return "srmap 0.1 (HIP, gfx950)"; }

int srmap_ctx_create(int device_id, srmap_ctx** out) {
  if (!out) return SRMAP_EINVAL;
  *out = nullptr;
  srmap_ctx* ctx = new (std::nothrow) srmap_ctx();
  if (!ctx) return SRMAP_ENOMEM;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) {
    fprintf(stderr, "srmap: no usable HIP device %d (%s); this library has no CPU path\n",
            device_id, e != hipSuccess ? hipGetErrorString(e) : "device index out of range");
    delete ctx;
    return SRMAP_EHIP;
  }
  ctx->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return SRMAP_EHIP;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
  *out = ctx;
  return SRMAP_OK;
}

void srmap_ctx_destroy(srmap_ctx* ctx) {
  if (!ctx) return;
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  for (int i = 0; i < 2; ++i) {
    if (ctx->h_stage[i]) (void)hipHostFree(ctx->h_stage[i]);
    if (ctx->h_event[i]) (void)hipEventDestroy(ctx->h_event[i]);
  }
  if (ctx->h_scal) (void)hipHostFree(ctx->h_scal);
  blas_release(ctx);
  delete ctx;
}

const char* srmap_last_error(const srmap_ctx* ctx) { return ctx ? ctx->error.c_str() : "null context"; }

int srmap_problem_create(srmap_ctx* ctx, const srmap_problem_desc* d, srmap_problem** out) {
  if (!ctx || !d || !out) return SRMAP_EINVAL;
  *out = nullptr;
  if (d->hr_width <= 0 || d->hr_height <= 0 || d->channels <= 0)
    return set_error(ctx, SRMAP_EINVAL, "image size and channel count must be positive");
  if (d->scale < 1) return set_error(ctx, SRMAP_EINVAL, "downsampling scale must be >= 1");  // image_model.cpp:66
  if (d->frames < 1) return set_error(ctx, SRMAP_EINVAL, "at least one frame is required");
  if (d->dtype != SRMAP_F64 && d->dtype != SRMAP_F32) return set_error(ctx, SRMAP_EINVAL, "bad dtype");
  if (d->hr_width > 32000 || d->hr_height > 32000)
    return set_error(ctx, SRMAP_EUNSUPPORTED, "image larger than warpAffine's 16-bit coordinates");
  const bool blur = d->blur_ksize > 0 && d->blur_sigma > 0.0;  // image_model.cpp:42
  if (blur && (d->blur_ksize % 2 != 1))
    return set_error(ctx, SRMAP_EINVAL, "blur radius must be an odd number");  // blur_module.cpp:18
  if (blur && d->blur_ksize * d->blur_ksize > kMaxBlurTaps)
    return set_error(ctx, SRMAP_EUNSUPPORTED, "blur kernel larger than %d taps", kMaxBlurTaps);
  if ((size_t)d->channels * d->hr_width * d->hr_height > (size_t)2147483647)
    return set_error(ctx, SRMAP_EINVAL, "number of data points exceeds INT_MAX");  // map_solver.cpp:96-101
  SRMAP_HIP(ctx, hipSetDevice(ctx->device));

  srmap_problem* p = new (std::nothrow) srmap_problem();
  if (!p) return SRMAP_ENOMEM;
  p->ctx = ctx;
  p->dtype = d->dtype;
  Geometry& g = p->geo;
  g.W = d->hr_width; g.H = d->hr_height; g.C = d->channels; g.K = d->frames; g.s = d->scale;
  const double scale_factor = 1.0 / (double)g.s;  // downsampling_module.cpp:24
  g.w = (int)(g.W * scale_factor);
  g.h = (int)(g.H * scale_factor);
  if (g.w <= 0 || g.h <= 0) { delete p; return set_error(ctx, SRMAP_EINVAL, "image smaller than the scale"); }
  g.b = blur ? d->blur_ksize : 1;
  g.hb = (g.b - 1) / 2;
  g.cr0 = 0; g.cr1 = g.H;
  g.rr0 = 0; g.rr1 = g.H;
  g.zlo = 0; g.zhi = 0;
  // Gaussian kernel: cv::getGaussianKernel (sigma > 0) and k * k^T, blur_module.cpp:20-22
  p->blur2d.assign((size_t)g.b * g.b, 1.0);
  p->blur1d.assign((size_t)g.b, 1.0);
  if (blur) {
    std::vector<double> k1(g.b);
    const double scale2x = -0.5 / (d->blur_sigma * d->blur_sigma);
    double sum = 0;
    for (int i = 0; i < g.b; ++i) {
      const double xx = i - (g.b - 1) * 0.5;
      k1[i] = std::exp(scale2x * xx * xx);
      sum += k1[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < g.b; ++i) k1[i] *= sum;
    p->blur1d = k1;
    for (int a = 0; a < g.b; ++a)
      for (int e = 0; e < g.b; ++e) p->blur2d[(size_t)a * g.b + e] = k1[a] * k1[e];
  }
  p->blur2d_t.resize(p->blur2d.size());
  for (int a = 0; a < g.b; ++a)
    for (int e = 0; e < g.b; ++e) p->blur2d_t[(size_t)a * g.b + e] = p->blur2d[(size_t)e * g.b + a];

  int rc = SRMAP_OK;
  p->has_motion = d->shifts_xy != nullptr;
  if (p->has_motion) {
    p->shifts.assign(d->shifts_xy, d->shifts_xy + 2 * (size_t)g.K);
    p->fwd_warps.resize(g.K);
    p->bwd_warps.resize(g.K);
    auto upload_ytab = [&](const std::vector<int>& t, WarpTaps<double>* wt) -> int {
      if (t.empty()) return SRMAP_OK;
      int* d = nullptr;
      if (hipMalloc((void**)&d, sizeof(int) * t.size()) != hipSuccess ||
          hipMemcpy(d, t.data(), sizeof(int) * t.size(), hipMemcpyHostToDevice) != hipSuccess)
        return set_error(ctx, SRMAP_ENOMEM, "hipMalloc failed");
      p->d_ytabs.push_back(d);
      wt->ytab = d;
      return SRMAP_OK;
    };
    std::vector<int> ytab;
    for (int k = 0; k < g.K && rc == SRMAP_OK; ++k) {
      rc = make_warp(ctx, g.W, g.H, p->shifts[2 * k], p->shifts[2 * k + 1], &p->fwd_warps[k], &ytab);
      if (rc == SRMAP_OK) rc = upload_ytab(ytab, &p->fwd_warps[k]);
      // the transpose warps an image of size (w*s, h*s)
      if (rc == SRMAP_OK)
        rc = make_warp(ctx, g.w * g.s, g.h * g.s, -p->shifts[2 * k], -p->shifts[2 * k + 1], &p->bwd_warps[k], &ytab);
      if (rc == SRMAP_OK) rc = upload_ytab(ytab, &p->bwd_warps[k]);
    }
  }
  std::vector<int> cmap, rmap;
  nearest_map(g.W, g.w, &cmap);
  nearest_map(g.H, g.h, &rmap);
  p->maps_regular = (g.W == g.w * g.s) && (g.H == g.h * g.s);
  for (int j = 0; j < g.w && p->maps_regular; ++j) p->maps_regular = cmap[j] == j * g.s;
  for (int i = 0; i < g.h && p->maps_regular; ++i) p->maps_regular = rmap[i] == i * g.s;
  if (p->maps_regular) {
    // NN upsampling used by the HR-resolution residual must replicate each LR
    // pixel exactly s*s times (objective_data_term.cpp:29, map_solver.cpp:80-85).
    std::vector<int> up;
    nearest_map(g.w, g.W, &up);
    for (int x = 0; x < g.W && p->maps_regular; ++x) p->maps_regular = up[x] == x / g.s;
    nearest_map(g.h, g.H, &up);
    for (int y = 0; y < g.H && p->maps_regular; ++y) p->maps_regular = up[y] == y / g.s;
  }
  auto fail = [&](int code) { srmap_problem_destroy(p); return code; };
  if (rc) return fail(rc);
  if (p->dtype == SRMAP_F32) {
    if (p->has_motion) { rc = upload_warps<float>(p, p->fwd_warps, &p->d_fwd_warps); if (rc) return fail(rc);
                         rc = upload_warps<float>(p, p->bwd_warps, &p->d_bwd_warps); if (rc) return fail(rc); }
    rc = upload_array<float>(p, p->blur2d, &p->d_blur); if (rc) return fail(rc);
    rc = upload_array<float>(p, p->blur2d_t, &p->d_blur_t); if (rc) return fail(rc);
  } else {
    if (p->has_motion) { rc = upload_warps<double>(p, p->fwd_warps, &p->d_fwd_warps); if (rc) return fail(rc);
                         rc = upload_warps<double>(p, p->bwd_warps, &p->d_bwd_warps); if (rc) return fail(rc); }
    rc = upload_array<double>(p, p->blur2d, &p->d_blur); if (rc) return fail(rc);
    rc = upload_array<double>(p, p->blur2d_t, &p->d_blur_t); if (rc) return fail(rc);
  }
  if (hipMalloc((void**)&p->d_col_map, sizeof(int) * g.w) != hipSuccess ||
      hipMalloc((void**)&p->d_row_map, sizeof(int) * g.h) != hipSuccess ||
      hipMalloc((void**)&p->d_cost, sizeof(double) * 8) != hipSuccess)
    return fail(set_error(ctx, SRMAP_ENOMEM, "hipMalloc failed"));
  (void)hipMemcpy(p->d_col_map, cmap.data(), sizeof(int) * g.w, hipMemcpyHostToDevice);
  (void)hipMemcpy(p->d_row_map, rmap.data(), sizeof(int) * g.h, hipMemcpyHostToDevice);
  // everything above used blocking copies; d_cost is first touched by kernels on caller streams: clear it on the
  // context's (non-blocking) stream and wait, so no legacy-stream work is left behind the creation
  if (hipMemsetAsync(p->d_cost, 0, sizeof(double) * 8, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
    return fail(set_error(ctx, SRMAP_EHIP, "clearing the cost scalars failed"));
  p->plan_gen++;
  if (ztile_plan(p)) ztile_preload(p);
  *out = p;
  return SRMAP_OK;
}

void srmap_problem_destroy(srmap_problem* p) {
  if (!p) return;
  ztile_release(p);
  void* bufs[] = {p->d_fwd_warps, p->d_bwd_warps, p->d_blur, p->d_blur_t, p->d_col_map, p->d_row_map,
                  p->d_obs, p->d_resid, p->d_regvals, p->d_x, p->d_g, p->d_tmp, p->d_partials, p->d_cost};
  for (void* b : bufs) if (b) (void)hipFree(b);
  for (int r = 0; r < p->nreg; ++r) if (p->reg[r].weights) (void)hipFree(p->reg[r].weights);
  for (int* t : p->d_ytabs) (void)hipFree(t);
  if (p->state_ev) (void)hipEventDestroy(p->state_ev);
  delete p;
}

int srmap_problem_set_impl(srmap_problem* p, int impl) {
  if (!p) return SRMAP_EINVAL;
  if (impl < SRMAP_IMPL_AUTO || impl > SRMAP_IMPL_TILED) return set_error(p->ctx, SRMAP_EINVAL, "bad impl");
  p->impl = impl;
  p->plan_gen++;
  return SRMAP_OK;
}

int srmap_problem_active_impl(const srmap_problem* p, int* impl) {
  if (!p || !impl) return SRMAP_EINVAL;
  const bool ztile = p->impl != SRMAP_IMPL_DIRECT && p->zplan != nullptr;
  *impl = ztile ? SRMAP_IMPL_TILED : SRMAP_IMPL_DIRECT;
  return SRMAP_OK;
}

int srmap_problem_set_cost_rows(srmap_problem* p, int hr_row0, int hr_row1) {
  if (!p) return SRMAP_EINVAL;
  const Geometry& g = p->geo;
  if (hr_row0 < 0 || hr_row1 > g.H || hr_row0 >= hr_row1)
    return set_error(p->ctx, SRMAP_EINVAL, "cost rows [%d, %d) outside the image", hr_row0, hr_row1);
  if (hr_row0 % g.s != 0 || (hr_row1 % g.s != 0 && hr_row1 != g.H))
    return set_error(p->ctx, SRMAP_EINVAL, "cost rows must be multiples of the scale %d", g.s);
  p->geo.cr0 = hr_row0;
  p->geo.cr1 = hr_row1;
  return SRMAP_OK;
}

int srmap_problem_lr_size(const srmap_problem* p, int* lw, int* lh) {
  if (!p) return SRMAP_EINVAL;
  if (lw) *lw = p->geo.w;
  if (lh) *lh = p->geo.h;
  return SRMAP_OK;
}

static int need_solver_geometry(srmap_problem* p) {
  if (!p->maps_regular)
    return set_error(p->ctx, SRMAP_EINVAL,
                     "HR size %dx%d is not LR size * scale (%d); MapSolver requires it (map_solver.cpp:72-76)",
                     p->geo.W, p->geo.H, p->geo.s);
  return SRMAP_OK;
}

int srmap_set_observations(srmap_problem* p, const double* lr_host) {
  if (!p || !lr_host) return SRMAP_EINVAL;
  int rc = need_solver_geometry(p);
  if (rc) return rc;
  SRMAP_HIP(p->ctx, hipSetDevice(p->ctx->device));
  rc = state_begin_write(p, p->ctx->stream);
  if (rc) return rc;
  rc = ensure(p, &p->d_obs, p->lr_count() * p->elem());
  if (rc) return rc;
  rc = convert_upload(p, lr_host, p->d_obs, p->lr_count(), p->ctx->stream);
  if (rc) return rc;
  p->have_obs = true;
  return SRMAP_OK;
}

int srmap_set_observations_device(srmap_problem* p, const void* lr_dev, void* hip_stream) {
  if (!p || !lr_dev) return SRMAP_EINVAL;
  int rc = need_solver_geometry(p);
  if (rc) return rc;
  SRMAP_HIP(p->ctx, hipSetDevice(p->ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : p->ctx->stream;
  rc = state_begin_write(p, st);
  if (rc) return rc;
  rc = ensure(p, &p->d_obs, p->lr_count() * p->elem());
  if (rc) return rc;
  SRMAP_HIP(p->ctx, hipMemcpyAsync(p->d_obs, lr_dev, p->lr_count() * p->elem(), hipMemcpyDeviceToDevice, st));
  SRMAP_HIP(p->ctx, hipStreamSynchronize(st));  // complete on return: lr_dev may be reused, any stream may evaluate
  p->have_obs = true;
  return SRMAP_OK;
}

int srmap_add_regularizer(srmap_problem* p, int kind, double lambda, int btv_range, double btv_decay,
                          int* reg_index) {
  if (!p) return SRMAP_EINVAL;
  if (kind != SRMAP_REG_TV && kind != SRMAP_REG_TV3D && kind != SRMAP_REG_BTV)
    return set_error(p->ctx, SRMAP_EINVAL, "unknown regularizer kind %d", kind);
  if (p->nreg >= kMaxRegularizers) return set_error(p->ctx, SRMAP_EUNSUPPORTED, "too many regularizers");
  RegSpec& rs = p->reg[p->nreg];
  rs = RegSpec{};
  rs.kind = kind;
  rs.lambda = lambda;
  if (kind == SRMAP_REG_BTV) {
    if (btv_range < 1) return set_error(p->ctx, SRMAP_EINVAL, "BTV range must be at least 1");  // btv_regularizer.cpp:58
    if (!(0 < btv_decay && btv_decay <= 1)) return set_error(p->ctx, SRMAP_EINVAL, "BTV decay must be in (0, 1]");
    if (btv_range > kMaxBtvRange) return set_error(p->ctx, SRMAP_EUNSUPPORTED, "BTV range > %d", kMaxBtvRange);
    rs.range = btv_range;
    rs.decay = btv_decay;
    for (int i = 0; i < 2 * kMaxBtvRange + 1; ++i) rs.pow_table[i] = std::pow(btv_decay, i);
  }
  if (reg_index) *reg_index = p->nreg;
  p->nreg++;
  p->plan_gen++;
  if (ztile_plan(p)) ztile_preload(p);
  return SRMAP_OK;
}

int srmap_clear_regularizers(srmap_problem* p) {
  if (!p) return SRMAP_EINVAL;
  for (int r = 0; r < p->nreg; ++r) if (p->reg[r].weights) { (void)hipFree(p->reg[r].weights); p->reg[r].weights = nullptr; }
  p->nreg = 0;
  p->plan_gen++;
  if (ztile_plan(p)) ztile_preload(p);
  return SRMAP_OK;
}

int srmap_set_irls_weights(srmap_problem* p, int reg, const double* w_host) {
  if (!p || reg < 0 || reg >= p->nreg) return SRMAP_EINVAL;
  SRMAP_HIP(p->ctx, hipSetDevice(p->ctx->device));
  RegSpec& rs = p->reg[reg];
  int rc = state_begin_write(p, p->ctx->stream);
  if (rc) return rc;
  if (!w_host) {
    if (rs.weights) { (void)hipFree(rs.weights); rs.weights = nullptr; }
    return SRMAP_OK;
  }
  rc = ensure(p, &rs.weights, p->hr_count() * p->elem());
  if (rc) return rc;
  return convert_upload(p, w_host, rs.weights, p->hr_count(), p->ctx->stream);
}

int srmap_update_irls_weights_device(srmap_problem* p, int reg, const void* x_dev, void* hip_stream) {
  if (!p || reg < 0 || reg >= p->nreg || !x_dev) return SRMAP_EINVAL;
  SRMAP_HIP(p->ctx, hipSetDevice(p->ctx->device));
  RegSpec& rs = p->reg[reg];
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : p->ctx->stream;
  int rc = state_begin_write(p, st);
  if (rc) return rc;
  rc = ensure(p, &rs.weights, p->hr_count() * p->elem());
  if (rc) return rc;
  if (p->dtype == SRMAP_F32) rc = launch_reg_weights<float>(p, p->geo, rs, (const float*)x_dev, (float*)rs.weights, st);
  else rc = launch_reg_weights<double>(p, p->geo, rs, (const double*)x_dev, (double*)rs.weights, st);
  if (rc) return rc;
  return state_end_write(p, st);  // asynchronous: evaluations on other streams wait for this event
}

// ---- operators on host buffers ----
int srmap_apply(srmap_problem* p, int frame, const double* hr, double* lr) {
  if (!p || !hr || !lr) return SRMAP_EINVAL;
  if (frame < 0 || frame >= p->geo.K) return set_error(p->ctx, SRMAP_EINVAL, "frame index %d out of range", frame);  // motion_shift.cpp:48-50
  SRMAP_HIP(p->ctx, hipSetDevice(p->ctx->device));
  hipStream_t st = p->ctx->stream;
  const Geometry& g = p->geo;
  const size_t nlr = (size_t)g.C * g.w * g.h;
  int rc = ensure(p, &p->d_x, p->hr_count() * p->elem());
  if (rc) return rc;
  rc = ensure(p, &p->d_tmp, p->hr_count() * p->elem());
  if (rc) return rc;
  rc = convert_upload(p, hr, p->d_x, p->hr_count(), st);
  if (rc) return rc;
  if (p->dtype == SRMAP_F32)
    rc = launch_forward_direct<float>(p, p->geo, (const float*)p->d_x, nullptr, p->geo.C, 0, (float*)p->d_tmp, frame, 1, nullptr, nullptr, st);
  else
    rc = launch_forward_direct<double>(p, p->geo, (const double*)p->d_x, nullptr, p->geo.C, 0, (double*)p->d_tmp, frame, 1, nullptr, nullptr, st);
  if (rc) return rc;
  return convert_download(p, p->d_tmp, lr, nlr, st);
}

int srmap_apply_transpose(srmap_problem* p, int frame, const double* lr, double* hr) {
  if (!p || !lr || !hr) return SRMAP_EINVAL;
  if (frame < 0 || frame >= p->geo.K) return set_error(p->ctx, SRMAP_EINVAL, "frame index %d out of range", frame);
  if (p->geo.W != p->geo.w * p->geo.s || p->geo.H != p->geo.h * p->geo.s)
    return set_error(p->ctx, SRMAP_EUNSUPPORTED, "transpose needs HR size == LR size * scale");
  SRMAP_HIP(p->ctx, hipSetDevice(p->ctx->device));
  hipStream_t st = p->ctx->stream;
  const Geometry& g = p->geo;
  const size_t nlr = (size_t)g.C * g.w * g.h;
  int rc = ensure(p, &p->d_g, p->hr_count() * p->elem());
  if (rc) return rc;
  rc = ensure(p, &p->d_tmp, p->hr_count() * p->elem());
  if (rc) return rc;
  rc = convert_upload(p, lr, p->d_tmp, nlr, st);
  if (rc) return rc;
  if (p->dtype == SRMAP_F32)
    rc = launch_gather_direct<float>(p, p->geo, (const float*)p->d_tmp, (float*)p->d_g, frame, 1, 1.0, false, st);
  else
    rc = launch_gather_direct<double>(p, p->geo, (const double*)p->d_tmp, (double*)p->d_g, frame, 1, 1.0, false, st);
  if (rc) return rc;
  return convert_download(p, p->d_g, hr, p->hr_count(), st);
}

int srmap_reg_values(srmap_problem* p, int reg, const double* x, double* values) {
  if (!p || !x || !values || reg < 0 || reg >= p->nreg) return SRMAP_EINVAL;
  SRMAP_HIP(p->ctx, hipSetDevice(p->ctx->device));
  hipStream_t st = p->ctx->stream;
  int rc = ensure(p, &p->d_x, p->hr_count() * p->elem());
  if (rc) return rc;
  rc = ensure(p, &p->d_regvals, p->hr_count() * p->elem());
  if (rc) return rc;
  rc = convert_upload(p, x, p->d_x, p->hr_count(), st);
  if (rc) return rc;
  if (p->dtype == SRMAP_F32)
    rc = launch_reg_values<float>(p, p->geo, p->reg[reg], (const float*)p->d_x, (float*)p->d_regvals, st);
  else
    rc = launch_reg_values<double>(p, p->geo, p->reg[reg], (const double*)p->d_x, (double*)p->d_regvals, st);
  if (rc) return rc;
  return convert_download(p, p->d_regvals, values, p->hr_count(), st);
}

int srmap_reg_values_and_gradient(srmap_problem* p, int reg, const double* x, const double* gc,
                                  double* values, double* gradient) {
  if (!p || !x || !gc || !values || !gradient || reg < 0 || reg >= p->nreg) return SRMAP_EINVAL;
  SRMAP_HIP(p->ctx, hipSetDevice(p->ctx->device));
  hipStream_t st = p->ctx->stream;
  const size_t n = p->hr_count();
  int rc = ensure(p, &p->d_x, n * p->elem()); if (rc) return rc;
  rc = ensure(p, &p->d_g, n * p->elem()); if (rc) return rc;
  rc = ensure(p, &p->d_tmp, n * p->elem()); if (rc) return rc;
  rc = ensure(p, &p->d_regvals, n * p->elem()); if (rc) return rc;
  rc = convert_upload(p, x, p->d_x, n, st); if (rc) return rc;
  rc = convert_upload(p, gc, p->d_tmp, n, st); if (rc) return rc;
  const RegSpec& rs = p->reg[reg];
  if (p->dtype == SRMAP_F32) {
    rc = launch_reg_values<float>(p, p->geo, rs, (const float*)p->d_x, (float*)p->d_regvals, st); if (rc) return rc;
    rc = launch_reg_gradient_direct<float>(p, p->geo, rs, (const float*)p->d_x, (const float*)p->d_tmp, 1.0,
                                           (const float*)p->d_regvals, (float*)p->d_g, false, nullptr, nullptr, st);
  } else {
    rc = launch_reg_values<double>(p, p->geo, rs, (const double*)p->d_x, (double*)p->d_regvals, st); if (rc) return rc;
    rc = launch_reg_gradient_direct<double>(p, p->geo, rs, (const double*)p->d_x, (const double*)p->d_tmp, 1.0,
                                            (const double*)p->d_regvals, (double*)p->d_g, false, nullptr, nullptr, st);
  }
  if (rc) return rc;
  rc = convert_download(p, p->d_regvals, values, n, st); if (rc) return rc;
  return convert_download(p, p->d_g, gradient, n, st);
}

// ---- objective ----
int srmap_eval_device(srmap_problem* p, unsigned terms, const void* x_dev, void* g_dev, double* cost,
                      void* hip_stream) {
  if (!p || !x_dev) return SRMAP_EINVAL;
  if (p->have_obs || !(terms & SRMAP_TERM_DATA)) { /* ok */ }
  SRMAP_HIP(p->ctx, hipSetDevice(p->ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : p->ctx->stream;
  int rc = eval_dispatch(p, terms, x_dev, g_dev, st);
  if (rc) return rc;
  if (cost) {
    SRMAP_HIP(p->ctx, hipMemcpyAsync(cost, p->d_cost, sizeof(double), hipMemcpyDeviceToHost, st));
    SRMAP_HIP(p->ctx, hipStreamSynchronize(st));
    if (*cost != *cost) {  // NaN: the input's, or the in-kernel reduction gave up waiting for a workgroup (sticky word)
      double flag = 0.0;
      SRMAP_HIP(p->ctx, hipMemcpy(&flag, p->d_cost + 6, sizeof(double), hipMemcpyDeviceToHost));
      if (flag != 0.0) {
        // the granules were left as they were (a late workgroup may still publish into them): re-initialise them
        // behind everything in flight, clear the word, and report -- the evaluation's gradient is not trustworthy
        SRMAP_HIP(p->ctx, hipDeviceSynchronize());
        ztile_rearm(p);
        SRMAP_HIP(p->ctx, hipMemset(p->d_cost + 6, 0, sizeof(double)));
        return set_error(p->ctx, SRMAP_EHIP, "in-kernel cost reduction timed out waiting for a workgroup (device fault or a wedged queue)");
      }
    }
  }
  return SRMAP_OK;
}

int srmap_last_cost(srmap_problem* p, double* cost) {
  if (!p || !cost) return SRMAP_EINVAL;
  SRMAP_HIP(p->ctx, hipSetDevice(p->ctx->device));
  SRMAP_HIP(p->ctx, hipDeviceSynchronize());
  SRMAP_HIP(p->ctx, hipMemcpy(cost, p->d_cost, sizeof(double), hipMemcpyDeviceToHost));
  return SRMAP_OK;
}

int srmap_eval(srmap_problem* p, unsigned terms, const double* x, double* cost, double* grad) {
  if (!p || !x) return SRMAP_EINVAL;  // CHECK_NOTNULL(estimated_image_data)
  SRMAP_HIP(p->ctx, hipSetDevice(p->ctx->device));
  hipStream_t st = p->ctx->stream;
  const size_t n = p->hr_count();
  int rc = ensure(p, &p->d_x, n * p->elem()); if (rc) return rc;
  if (grad) { rc = ensure(p, &p->d_g, n * p->elem()); if (rc) return rc; }
  rc = convert_upload(p, x, p->d_x, n, st); if (rc) return rc;
  double c = 0;
  rc = srmap_eval_device(p, terms, p->d_x, grad ? p->d_g : nullptr, &c, st); if (rc) return rc;
  if (cost) *cost = c;
  if (grad) return convert_download(p, p->d_g, grad, n, st);
  return SRMAP_OK;
}

int srmap_device_alloc(srmap_ctx* ctx, size_t bytes, void** dev) {
  if (!ctx || !dev) return SRMAP_EINVAL;
  SRMAP_HIP(ctx, hipSetDevice(ctx->device));
  SRMAP_HIP(ctx, hipMalloc(dev, bytes ? bytes : 8));
  return SRMAP_OK;
}
int srmap_device_free(srmap_ctx* ctx, void* dev) {
  if (!ctx) return SRMAP_EINVAL;
  if (dev) SRMAP_HIP(ctx, hipFree(dev));
  return SRMAP_OK;
}
int srmap_upload(srmap_problem* p, const double* host, void* dev, size_t count) {
  if (!p || !host || !dev) return SRMAP_EINVAL;
  SRMAP_HIP(p->ctx, hipSetDevice(p->ctx->device));
  return convert_upload(p, host, dev, count, p->ctx->stream);
}
int srmap_download(srmap_problem* p, const void* dev, double* host, size_t count) {
  if (!p || !host || !dev) return SRMAP_EINVAL;
  SRMAP_HIP(p->ctx, hipSetDevice(p->ctx->device));
  return convert_download(p, dev, host, count, p->ctx->stream);
}
int srmap_synchronize(srmap_ctx* ctx) {
  if (!ctx) return SRMAP_EINVAL;
  SRMAP_HIP(ctx, hipSetDevice(ctx->device));
  SRMAP_HIP(ctx, hipDeviceSynchronize());
  return SRMAP_OK;
}

void srmap_irls_options_default(srmap_irls_options* o) {
  if (!o) return;
  o->struct_size = (int)sizeof(srmap_irls_options);
  o->max_num_solver_iterations = 50;
  o->gradient_norm_threshold = 1.0e-6;
  o->cost_decrease_threshold = 1.0e-6;
  o->parameter_variation_threshold = 1.0e-6;
  o->split_channels = 0;
  o->max_num_irls_iterations = 20;
  o->irls_cost_difference_threshold = 1.0e-5;
  o->host_paced_passes = 0;
}

int srmap_solve_sharded(srmap_problem* p, srmap_comm* comm, const srmap_shard_desc* shard,
                        const srmap_irls_options* options, const double* x0, double* x_out,
                        srmap_solve_report* report) {
  if (!p || !x0 || !x_out) return SRMAP_EINVAL;
  srmap_irls_options o;
  if (options) {
    // the first member is the size of the struct the caller was compiled with: another layout is refused, not guessed at
    if (options->struct_size != (int)sizeof(srmap_irls_options))
      return set_error(p->ctx, SRMAP_EINVAL, "srmap_irls_options: struct_size %d, this library expects %d (fill the struct with "
                       "srmap_irls_options_default() of the matching include/srmap.h)", options->struct_size, (int)sizeof(srmap_irls_options));
    o = *options;
  } else {
    srmap_irls_options_default(&o);
  }
  SRMAP_HIP(p->ctx, hipSetDevice(p->ctx->device));
  return solve_impl(p, comm, shard, &o, x0, x_out, report);
}

int srmap_problem_selfcheck(const srmap_problem* p, double* beta_denominator_rel_dev) {
  if (!p || !beta_denominator_rel_dev) return SRMAP_EINVAL;
  *beta_denominator_rel_dev = p->selfcheck_beta_den;
  return SRMAP_OK;
}

int srmap_solve(srmap_problem* p, const srmap_irls_options* options, const double* x0, double* x_out,
                srmap_solve_report* report) {
  return srmap_solve_sharded(p, nullptr, nullptr, options, x0, x_out, report);
}

}  // extern "C"
