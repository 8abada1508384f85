/*
 * srmap_oracle.c -- CPU restatement of the reference MAP super-resolution
 * gradient path.  TEST INFRASTRUCTURE ONLY (see srmap_oracle.h): loaded by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
 * the product path.
 *
 * Written to be read next to the reference: every function names the
 * reference lines it follows.  Third-party arithmetic that the reference
 * delegates to OpenCV (3.2+, un-pinned, absent from /root/reference) is
 * restated from OpenCV 3.x imgproc's published behaviour and anchored on the
 * reference's own call sites and test literals.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (no FMA contraction so that
 * the CG trajectory can be compared bit-for-bit with the vendored ALGLIB).
 */
#include "srmap_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* util::GetPixelIndex, src/util/util.cpp:81-89. */
static inline long pix(int W, int H, int c, int row, int col) {
  return (long)c * W * H + (long)row * W + col;
}

static void* xmalloc(size_t n) {
  void* p = malloc(n ? n : 1);
  if (!p) abort();
  return p;
}

/* cvRound / saturate_cast<int>(double): round to nearest, ties to even. */
static inline int cv_round(double v) { return (int)lrint(v); }
static inline int cv_floor(double v) { return (int)floor(v); }
static inline int sat_short(int v) {
  return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}

/* ===================================================================== */
/* OpenCV restatements                                                   */
/* ===================================================================== */

/* cv::warpAffine without WARP_INVERSE_MAP first inverts M = [1 0 dx; 0 1 dy]
 * (general 2x3 inversion formula kept so that signed zeros come out as in
 * OpenCV), then walks destination pixels with AB_BITS = 10 fixed-point
 * coordinates rounded to INTER_BITS = 5 fractional bits. */
void sro_warp_tables(int W, int H, double dx, double dy, int* X, int* Y) {
  double M[6] = {1.0, 0.0, dx, 0.0, 1.0, dy};
  double D = M[0] * M[4] - M[1] * M[3];
  D = D != 0 ? 1.0 / D : 0;
  double A11 = M[4] * D, A22 = M[0] * D;
  M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
  double b1 = -M[0] * M[2] - M[1] * M[5];
  double b2 = -M[3] * M[2] - M[4] * M[5];
  M[2] = b1; M[5] = b2;
  const int AB_SCALE = 1 << 10;
  const int round_delta = AB_SCALE / 32 / 2;  /* 16 */
  /* For a pure shift bdelta[x] = M[3]*x == 0 and M[1]*y == 0, so the x table
   * does not depend on y and the y table does not depend on x. */
  for (int x = 0; x < W; ++x) {
    int adelta = cv_round(M[0] * x * AB_SCALE);
    int X0 = cv_round((M[1] * 0 + M[2]) * AB_SCALE) + round_delta;
    X[x] = (X0 + adelta) >> 5;
  }
  for (int y = 0; y < H; ++y) {
    int Y0 = cv_round((M[4] * y + M[5]) * AB_SCALE) + round_delta;
    int bdelta = cv_round(M[3] * 0 * AB_SCALE);
    Y[y] = (Y0 + bdelta) >> 5;
  }
}

void sro_warp_shift(const double* src, double* dst, int W, int H,
                    double dx, double dy) {
  int* X = (int*)xmalloc(sizeof(int) * W);
  int* Y = (int*)xmalloc(sizeof(int) * H);
  sro_warp_tables(W, H, dx, dy, X, Y);
  /* remapBilinear<double>: float32 weight table, double accumulation,
   * BORDER_CONSTANT 0. */
  for (int y = 0; y < H; ++y) {
    const int sy = sat_short(Y[y] >> 5);
    const int fy = Y[y] & 31;
    for (int x = 0; x < W; ++x) {
      const int sx = sat_short(X[x] >> 5);
      const int fx = X[x] & 31;
      const float tx1 = (float)fx * (1.f / 32), tx0 = 1.f - tx1;
      const float ty1 = (float)fy * (1.f / 32), ty0 = 1.f - ty1;
      const float w[4] = {ty0 * tx0, ty0 * tx1, ty1 * tx0, ty1 * tx1};
      double out;
      if ((unsigned)sx < (unsigned)(W - 1 > 0 ? W - 1 : 0) &&
          (unsigned)sy < (unsigned)(H - 1 > 0 ? H - 1 : 0)) {
        const double* S = src + (long)sy * W + sx;
        out = S[0] * w[0] + S[1] * w[1] + S[W] * w[2] + S[W + 1] * w[3];
      } else if (sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0) {
        out = 0;
      } else {
        const int x0 = sx, x1 = sx + 1, y0 = sy, y1 = sy + 1;
        const double v0 = (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H)
                              ? src[(long)y0 * W + x0] : 0;
        const double v1 = (x1 >= 0 && x1 < W && y0 >= 0 && y0 < H)
                              ? src[(long)y0 * W + x1] : 0;
        const double v2 = (x0 >= 0 && x0 < W && y1 >= 0 && y1 < H)
                              ? src[(long)y1 * W + x0] : 0;
        const double v3 = (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H)
                              ? src[(long)y1 * W + x1] : 0;
        out = v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
      }
      dst[(long)y * W + x] = out;
    }
  }
  free(X);
  free(Y);
}

void sro_gaussian_kernel(int ksize, double sigma, double* out1d,
                         double* out2d) {
  /* getGaussianKernel, sigma > 0 branch: t_i = exp(-x_i^2 / (2 sigma^2)),
   * scaled by the reciprocal of the sum. */
  const double scale2x = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < ksize; ++i) {
    const double x = i - (ksize - 1) * 0.5;
    const double t = exp(scale2x * x * x);
    out1d[i] = t;
    sum += t;
  }
  sum = 1.0 / sum;
  for (int i = 0; i < ksize; ++i) out1d[i] *= sum;
  /* blur_kernel_ = kernel_x * kernel_y.t()  (blur_module.cpp:20-22) */
  if (out2d)
    for (int a = 0; a < ksize; ++a)
      for (int e = 0; e < ksize; ++e) out2d[a * ksize + e] = out1d[a] * out1d[e];
}

void sro_filter2d(const double* src, double* dst, int W, int H,
                  const double* kernel, int kw, int kh) {
  const int ax = kw / 2, ay = kh / 2;
  for (int r = 0; r < H; ++r)
    for (int c = 0; c < W; ++c) {
      double s = 0;
      for (int a = 0; a < kh; ++a) {
        const int rr = r + a - ay;
        for (int e = 0; e < kw; ++e) {
          const double kf = kernel[a * kw + e];
          if (kf == 0) continue;  /* OpenCV keeps only non-zero taps */
          const int cc = c + e - ax;
          const double v =
              (rr >= 0 && rr < H && cc >= 0 && cc < W) ? src[(long)rr * W + cc] : 0;
          s += kf * v;
        }
      }
      dst[(long)r * W + c] = s;
    }
}

void sro_nearest_map(int src_len, int dst_len, int* map) {
  /* cv::resize: inv_scale = dsize/ssize; resizeNN: ifx = 1/inv_scale,
   * sx = min(cvFloor(x*ifx), ssize-1). */
  const double inv_scale = (double)dst_len / src_len;
  const double ifx = 1.0 / inv_scale;
  for (int x = 0; x < dst_len; ++x) {
    int sx = cv_floor(x * ifx);
    map[x] = sx < src_len - 1 ? sx : src_len - 1;
  }
}

void sro_resize_nearest(const double* src, int sw, int sh, double* dst, int dw,
                        int dh) {
  int* mx = (int*)xmalloc(sizeof(int) * dw);
  int* my = (int*)xmalloc(sizeof(int) * dh);
  sro_nearest_map(sw, dw, mx);
  sro_nearest_map(sh, dh, my);
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x)
      dst[(long)y * dw + x] = src[(long)my[y] * sw + mx[x]];
  free(mx);
  free(my);
}

/* ===================================================================== */
/* Reference's own image code                                            */
/* ===================================================================== */

void sro_resize_additive(const double* src, int sw, int sh, double* dst, int dw,
                         int dh) {
  /* image_data.cpp:80-134 */
  memset(dst, 0, sizeof(double) * (size_t)dw * dh);
  const int upsample = sw <= dw && sh <= dh;
  if (upsample) {
    const int ys = dh / sh, xs = dw / sw;
    for (int row = 0; row < sh; ++row)
      for (int col = 0; col < sw; ++col)
        dst[(long)(row * ys) * dw + col * xs] = src[(long)row * sw + col];
  } else {
    const int ys = sh / dh, xs = sw / dw;
    for (int row = 0; row < sh; ++row)
      for (int col = 0; col < sw; ++col)
        dst[(long)(row / ys) * dw + col / xs] += src[(long)row * sw + col];
  }
}

int sro_downsampled_len(int len, int scale) {
  const double scale_factor = 1.0 / (double)scale;
  return (int)(len * scale_factor);
}

static int model_has_blur(const sro_model* m) {
  /* image_model.cpp:42-46 */
  return m->blur_ksize > 0 && m->blur_sigma > 0.0;
}

void sro_model_apply(const sro_model* m, int k, const double* hr, int W, int H,
                     int C, double* lr) {
  const int w = sro_downsampled_len(W, m->scale);
  const int h = sro_downsampled_len(H, m->scale);
  const long N = (long)W * H;
  double* a = (double*)xmalloc(sizeof(double) * N);
  double* b = (double*)xmalloc(sizeof(double) * N);
  double* k1 = NULL;
  double* k2 = NULL;
  if (model_has_blur(m)) {
    k1 = (double*)xmalloc(sizeof(double) * m->blur_ksize);
    k2 = (double*)xmalloc(sizeof(double) * m->blur_ksize * m->blur_ksize);
    sro_gaussian_kernel(m->blur_ksize, m->blur_sigma, k1, k2);
  }
  for (int c = 0; c < C; ++c) {
    memcpy(a, hr + c * N, sizeof(double) * N);
    if (m->num_frames > 0) { /* MotionModule::ApplyToImage */
      sro_warp_shift(a, b, W, H, m->shifts[2 * k], m->shifts[2 * k + 1]);
      double* t = a; a = b; b = t;
    }
    if (k2) { /* BlurModule::ApplyToImage */
      sro_filter2d(a, b, W, H, k2, m->blur_ksize, m->blur_ksize);
      double* t = a; a = b; b = t;
    }
    /* DownsamplingModule::ApplyToImage */
    sro_resize_nearest(a, W, H, lr + (long)c * w * h, w, h);
  }
  free(a); free(b); free(k1); free(k2);
}

void sro_model_apply_transpose(const sro_model* m, int k, const double* lr,
                               int w, int h, int C, double* hr) {
  /* DownsamplingModule::ApplyTransposeToImage: ResizeImage(scale, ADDITIVE)
   * -> new size (int)(w * scale). */
  const int W = (int)(w * (double)m->scale);
  const int H = (int)(h * (double)m->scale);
  const long N = (long)W * H;
  double* a = (double*)xmalloc(sizeof(double) * N);
  double* b = (double*)xmalloc(sizeof(double) * N);
  double* k1 = NULL;
  double* k2 = NULL;
  double* k2t = NULL;
  if (model_has_blur(m)) {
    const int ks = m->blur_ksize;
    k1 = (double*)xmalloc(sizeof(double) * ks);
    k2 = (double*)xmalloc(sizeof(double) * ks * ks);
    k2t = (double*)xmalloc(sizeof(double) * ks * ks);
    sro_gaussian_kernel(ks, m->blur_sigma, k1, k2);
    for (int i = 0; i < ks; ++i) /* blur_kernel_.t(), blur_module.cpp:35 */
      for (int j = 0; j < ks; ++j) k2t[i * ks + j] = k2[j * ks + i];
  }
  for (int c = 0; c < C; ++c) {
    sro_resize_additive(lr + (long)c * w * h, w, h, a, W, H);
    if (k2t) {
      sro_filter2d(a, b, W, H, k2t, m->blur_ksize, m->blur_ksize);
      double* t = a; a = b; b = t;
    }
    if (m->num_frames > 0) { /* MotionModule::ApplyTransposeToImage */
      sro_warp_shift(a, b, W, H, -m->shifts[2 * k], -m->shifts[2 * k + 1]);
      double* t = a; a = b; b = t;
    }
    memcpy(hr + c * N, a, sizeof(double) * N);
  }
  free(a); free(b); free(k1); free(k2); free(k2t);
}

/* ===================================================================== */
/* Regularizers                                                          */
/* ===================================================================== */

/* tv_regularizer.cpp:21-36 */
static double tv_xgrad(const double* x, int W, int H, int c, int row, int col) {
  if (col >= 0 && col + 1 < W)
    return x[pix(W, H, c, row, col + 1)] - x[pix(W, H, c, row, col)];
  return 0;
}
/* tv_regularizer.cpp:40-55 */
static double tv_ygrad(const double* x, int W, int H, int c, int row, int col) {
  if (row >= 0 && row + 1 < H)
    return x[pix(W, H, c, row + 1, col)] - x[pix(W, H, c, row, col)];
  return 0;
}
/* tv_regularizer.cpp:60-71 */
static double tv_zgrad(const double* x, int W, int H, int c, int row, int col) {
  return x[pix(W, H, c + 1, row, col)] - x[pix(W, H, c, row, col)];
}
/* tv_regularizer.cpp:75-87 */
static double tv_abs(const double* x, int W, int H, int c, int row, int col) {
  const double yv = fabs(tv_ygrad(x, W, H, c, row, col));
  const double xv = fabs(tv_xgrad(x, W, H, c, row, col));
  return yv + xv;
}
/* tv_regularizer.cpp:92-106 */
static double tv_3d(const double* x, int W, int H, int C, int c, int row,
                    int col) {
  double tv = tv_abs(x, W, H, c, row, col);
  if (c + 1 < C) tv += fabs(tv_zgrad(x, W, H, c, row, col));
  return tv;
}

/* btv_regularizer.cpp:19-46 */
static double btv_value(const double* x, int W, int H, int c, int row, int col,
                        int range, double decay) {
  double tv = 0.0;
  const long index = pix(W, H, c, row, col);
  for (int i = 0; i <= range; ++i)
    for (int j = 0; j <= range; ++j) {
      const int orow = row + i, ocol = col + j;
      if (orow >= H || ocol >= W) continue;
      const long oindex = pix(W, H, c, orow, ocol);
      const double d = pow(decay, i + j);
      tv += d * fabs(x[index] - x[oindex]);
    }
  return tv;
}

void sro_reg_values(const sro_regularizer* r, const double* x, int W, int H,
                    int C, double* values) {
  for (int c = 0; c < C; ++c)
    for (int row = 0; row < H; ++row)
      for (int col = 0; col < W; ++col) {
        const long index = pix(W, H, c, row, col);
        if (r->kind == SRO_REG_BTV)
          values[index] =
              btv_value(x, W, H, c, row, col, r->btv_range, r->btv_decay);
        else if (r->kind == SRO_REG_TV3D)
          values[index] = tv_3d(x, W, H, C, c, row, col);
        else
          values[index] = tv_abs(x, W, H, c, row, col);
      }
}

static void tv_gradient(int use3d, const double* x, const double* gc,
                        const double* res, int W, int H, int C, double* g) {
  /* tv_regularizer.cpp:143-224 */
  for (int c = 0; c < C; ++c)
    for (int row = 0; row < H; ++row)
      for (int col = 0; col < W; ++col) {
        const long index = pix(W, H, c, row, col);
        double didi = 0.0;
        const double xg = tv_xgrad(x, W, H, c, row, col);
        if (xg < 0.0) didi += 1.0; else if (xg > 0.0) didi -= 1.0;
        const double yg = tv_ygrad(x, W, H, c, row, col);
        if (yg < 0.0) didi += 1.0; else if (yg > 0.0) didi -= 1.0;
        /* NOTE: no z term in the self derivative even in 3-D mode
         * (tv_regularizer.cpp:154-170) -- reproduced as is. */
        g[index] += 2 * gc[index] * res[index] * didi;
        if (col - 1 >= 0) {
          const long li = pix(W, H, c, row, col - 1);
          const double lg = tv_xgrad(x, W, H, c, row, col - 1);
          double d = 0.0;
          if (lg > 0.0) d = 1.0; else if (lg < 0.0) d = -1.0;
          g[index] += 2 * gc[li] * res[li] * d;
        }
        if (row - 1 >= 0) {
          const long ai = pix(W, H, c, row - 1, col);
          const double ag = tv_ygrad(x, W, H, c, row - 1, col);
          double d = 0.0;
          if (ag > 0.0) d = 1.0; else if (ag < 0.0) d = -1.0;
          g[index] += 2 * gc[ai] * res[ai] * d;
        }
        if (use3d && c > 0) {
          const long bi = pix(W, H, c - 1, row, col);
          const double bg = tv_zgrad(x, W, H, c - 1, row, col);
          double d = 0.0;
          if (bg > 0.0) d = 1.0; else if (bg < 0.0) d = -1.0;
          g[index] += 2 * gc[bi] * res[bi] * d;
        }
      }
}

static void btv_gradient(int range, double decay, const double* x,
                         const double* gc, const double* res, int W, int H,
                         int C, double* g) {
  /* btv_regularizer.cpp:105-166 */
  for (int c = 0; c < C; ++c)
    for (int row = 0; row < H; ++row)
      for (int col = 0; col < W; ++col) {
        const long index = pix(W, H, c, row, col);
        double didi = 0.0;
        for (int i = 0; i < range; ++i)      /* exclusive range here */
          for (int j = 0; j < range; ++j) {
            const int orow = row + i, ocol = col + j;
            if (orow >= H || ocol >= W) continue;
            const long oi = pix(W, H, c, orow, ocol);
            const double diff = x[index] - x[oi];
            double ag = 0.0;
            if (diff > 0.0) ag = 1.0; else if (diff < 0.0) ag = -1.0;
            didi += pow(decay, i + j) * ag;
          }
        g[index] += 2 * gc[index] * res[index] * didi;
        for (int i = 0; i < range; ++i)
          for (int j = 0; j < range; ++j) {
            const int orow = row - i, ocol = col - j;
            /* compares COORDINATES with 0, not offsets: the absolute pixel
             * (0,0) never back-propagates (btv_regularizer.cpp:143-146). */
            if ((orow == 0 && ocol == 0) || orow < 0 || ocol < 0) continue;
            const long oi = pix(W, H, c, orow, ocol);
            const double diff = x[oi] - x[index];
            double didj = 0.0;
            if (diff < 0.0) didj = 1.0; else if (diff > 0.0) didj = -1.0;
            didj *= pow(decay, i + j);
            g[index] += 2 * gc[oi] * res[oi] * didj;
          }
      }
}

void sro_reg_values_and_gradient(const sro_regularizer* r, const double* x,
                                 const double* gc, int W, int H, int C,
                                 double* values, double* gradient) {
  sro_reg_values(r, x, W, H, C, values);
  memset(gradient, 0, sizeof(double) * (size_t)W * H * C);
  if (r->kind == SRO_REG_BTV)
    btv_gradient(r->btv_range, r->btv_decay, x, gc, values, W, H, C, gradient);
  else
    tv_gradient(r->kind == SRO_REG_TV3D, x, gc, values, W, H, C, gradient);
}

/* ===================================================================== */
/* MAP problem / objective                                               */
/* ===================================================================== */

#define SRO_MAX_REG 8
struct sro_problem {
  sro_model model;
  double* shifts;
  int W, H, C, w, h, K;
  double* obs_hr; /* observations_: [K][C][H][W], NN-upsampled */
  int nreg;
  sro_regularizer reg[SRO_MAX_REG];
  double lambda[SRO_MAX_REG];
  double* weights[SRO_MAX_REG];
};

sro_problem* sro_problem_create(const sro_model* m, const double* lr_frames,
                                int w, int h, int C) {
  /* K observations are required even when the chain has no MotionModule;
   * num_frames carries K and shifts == NULL means "no MotionModule". */
  sro_problem* p = (sro_problem*)xmalloc(sizeof(*p));
  memset(p, 0, sizeof(*p));
  p->model = *m;
  p->K = m->num_frames;
  if (m->shifts) {
    p->shifts = (double*)xmalloc(sizeof(double) * 2 * p->K);
    memcpy(p->shifts, m->shifts, sizeof(double) * 2 * p->K);
    p->model.shifts = p->shifts;
  } else {
    p->model.num_frames = 0;
  }
  p->w = w; p->h = h; p->C = C;
  p->W = w * m->scale; p->H = h * m->scale; /* map_solver.cpp:72-76 */
  const long N = (long)p->W * p->H;
  p->obs_hr = (double*)xmalloc(sizeof(double) * N * C * p->K);
  for (int k = 0; k < p->K; ++k)   /* map_solver.cpp:80-85 */
    for (int c = 0; c < C; ++c)
      sro_resize_nearest(lr_frames + ((long)k * C + c) * w * h, w, h,
                         p->obs_hr + ((long)k * C + c) * N, p->W, p->H);
  return p;
}

void sro_problem_destroy(sro_problem* p) {
  if (!p) return;
  for (int i = 0; i < p->nreg; ++i) free(p->weights[i]);
  free(p->obs_hr);
  free(p->shifts);
  free(p);
}

int sro_problem_hr_width(const sro_problem* p) { return p->W; }
int sro_problem_hr_height(const sro_problem* p) { return p->H; }

int sro_problem_add_regularizer(sro_problem* p, const sro_regularizer* r,
                                double lambda) {
  if (p->nreg >= SRO_MAX_REG) return -1;
  const long n = (long)p->W * p->H * p->C;
  p->reg[p->nreg] = *r;
  p->lambda[p->nreg] = lambda;
  p->weights[p->nreg] = (double*)xmalloc(sizeof(double) * n);
  for (long i = 0; i < n; ++i) p->weights[p->nreg][i] = 1.0;
  return p->nreg++;
}

void sro_problem_set_irls_weights(sro_problem* p, int reg, const double* w) {
  const long n = (long)p->W * p->H * p->C;
  if (w) memcpy(p->weights[reg], w, sizeof(double) * n);
  else for (long i = 0; i < n; ++i) p->weights[reg][i] = 1.0;
}

/* ComputeTermForObservation, objective_data_term.cpp:15-75, for the channel
 * range [c0, c1) of the observation and x holding (c1-c0) channels. */
static double data_term_one(const sro_problem* p, int k, int c0, int c1,
                            const double* x, double* gradient) {
  const int W = p->W, H = p->H, Cn = c1 - c0, s = p->model.scale;
  const long N = (long)W * H;
  const int w = sro_downsampled_len(W, s), h = sro_downsampled_len(H, s);
  double* lr = (double*)xmalloc(sizeof(double) * (size_t)w * h * Cn);
  double* up = (double*)xmalloc(sizeof(double) * N * Cn);
  sro_model_apply(&p->model, k, x, W, H, Cn, lr);               /* :27-28 */
  for (int c = 0; c < Cn; ++c)                                   /* :29 */
    sro_resize_nearest(lr + (long)c * w * h, w, h, up + c * N, W, H);
  double sum = 0;
  double* residuals = (double*)xmalloc(sizeof(double) * N * Cn);
  for (int c = 0; c < Cn; ++c) {                                 /* :36-50 */
    const double* d = up + c * N;
    const double* o = p->obs_hr + ((long)k * p->C + c + c0) * N;
    for (long i = 0; i < N; ++i) {
      const double r = d[i] - o[i];
      residuals[c * N + i] = r;
      sum += r * r;
    }
  }
  if (gradient) {                                                /* :54-72 */
    const int lw = W / s, lh = H / s;
    double* rl = (double*)xmalloc(sizeof(double) * (size_t)lw * lh * Cn);
    for (int c = 0; c < Cn; ++c)
      sro_resize_additive(residuals + c * N, W, H, rl + (long)c * lw * lh, lw,
                          lh);
    sro_model_apply_transpose(&p->model, k, rl, lw, lh, Cn, up);
    for (int c = 0; c < Cn; ++c)
      for (long i = 0; i < N; ++i) gradient[c * N + i] += 2 * up[c * N + i];
    free(rl);
  }
  free(residuals); free(up); free(lr);
  return sum;
}

static double data_term_range(const sro_problem* p, int c0, int c1,
                              const double* x, double* gradient) {
  double sum = 0.0;  /* objective_data_term.cpp:98-116 */
  for (int k = 0; k < p->K; ++k) sum += data_term_one(p, k, c0, c1, x, gradient);
  return sum;
}

double sro_data_term(const sro_problem* p, const double* x, double* gradient) {
  return data_term_range(p, 0, p->C, x, gradient);
}

static double irls_term_range(const sro_problem* p, int reg, int Cn,
                              const double* weights, const double* x,
                              double* gradient) {
  /* objective_irls_regularization_term.cpp:10-58 */
  const double lambda = p->lambda[reg];
  if (lambda <= 0.0) return 0.0;
  const long n = (long)p->W * p->H * Cn;
  double* gc = (double*)xmalloc(sizeof(double) * n);
  double* vals = (double*)xmalloc(sizeof(double) * n);
  double* part = (double*)xmalloc(sizeof(double) * n);
  for (long i = 0; i < n; ++i) gc[i] = lambda * weights[i];
  sro_reg_values_and_gradient(&p->reg[reg], x, gc, p->W, p->H, Cn, vals, part);
  double sum = 0.0;
  for (long i = 0; i < n; ++i) {
    const double r = vals[i], wt = weights[i];
    sum += lambda * wt * r * r;
    if (gradient) gradient[i] += part[i];
  }
  free(gc); free(vals); free(part);
  return sum;
}

double sro_irls_reg_term(const sro_problem* p, int reg, const double* x,
                         double* gradient) {
  return irls_term_range(p, reg, p->C, p->weights[reg], x, gradient);
}

double sro_objective(const sro_problem* p, const double* x, double* gradient) {
  /* objective_function.cpp:5-20 */
  const long n = (long)p->W * p->H * p->C;
  if (gradient) for (long i = 0; i < n; ++i) gradient[i] = 0.0;
  double sum = 0.0;
  sum += sro_data_term(p, x, gradient);
  for (int r = 0; r < p->nreg; ++r) sum += sro_irls_reg_term(p, r, x, gradient);
  return sum;
}

/* ===================================================================== */
/* Nonlinear CG (ALGLIB mincg, default settings) restated                */
/* ===================================================================== */

/* ae_v_dotproduct (ap.cpp:4667-4692): groups of four. */
static double vdot(const double* a, const double* b, long n) {
  double r = 0;
  long n4 = n / 4, left = n % 4, i;
  for (i = 0; i < n4; ++i, a += 4, b += 4)
    r += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  for (i = 0; i < left; ++i, ++a, ++b) r += a[0] * b[0];
  return r;
}

static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }

/* More'-Thuente safeguarded step (MINPACK-2 dcstep as ALGLIB's linmin_mcstep
 * uses it, alglibinternal.cpp:12972-13232): updates the bracket
 * [stx, sty] and proposes the next trial step. */
typedef struct {
  double stx, fx, dx, sty, fy, dy;
} mt_bracket;

static double mt_cubic_gamma(double theta, double da, double db, int clamp0) {
  const double s = dmax(fabs(theta), dmax(fabs(da), fabs(db)));
  double t = (theta / s) * (theta / s) - da / s * (db / s);
  if (clamp0) t = dmax(0.0, t);
  return s * sqrt(t);
}

static void mt_step(mt_bracket* b, double* stp, double fp, double dp,
                    int* brackt, double stmin, double stmax, int* info) {
  *info = 0;
  if ((*brackt && (*stp <= dmin(b->stx, b->sty) || *stp >= dmax(b->stx, b->sty))) ||
      b->dx * (*stp - b->stx) >= 0 || stmax < stmin)
    return;
  const double sgnd = dp * (b->dx / fabs(b->dx));
  int bound;
  double stpf;
  if (fp > b->fx) {
    /* higher value: minimum bracketed; cubic vs quadratic, closer to stx */
    *info = 1;
    bound = 1;
    const double theta = 3 * (b->fx - fp) / (*stp - b->stx) + b->dx + dp;
    double gamma = mt_cubic_gamma(theta, b->dx, dp, 0);
    if (*stp < b->stx) gamma = -gamma;
    const double p = gamma - b->dx + theta;
    const double q = gamma - b->dx + gamma + dp;
    const double r = p / q;
    const double stpc = b->stx + r * (*stp - b->stx);
    const double stpq =
        b->stx + b->dx / ((b->fx - fp) / (*stp - b->stx) + b->dx) / 2 * (*stp - b->stx);
    if (fabs(stpc - b->stx) < fabs(stpq - b->stx)) stpf = stpc;
    else stpf = stpc + (stpq - stpc) / 2;
    *brackt = 1;
  } else if (sgnd < 0) {
    /* lower value, derivatives of opposite sign: bracketed; cubic vs secant,
     * farther from stp */
    *info = 2;
    bound = 0;
    const double theta = 3 * (b->fx - fp) / (*stp - b->stx) + b->dx + dp;
    double gamma = mt_cubic_gamma(theta, b->dx, dp, 0);
    if (*stp > b->stx) gamma = -gamma;
    const double p = gamma - dp + theta;
    const double q = gamma - dp + gamma + b->dx;
    const double r = p / q;
    const double stpc = *stp + r * (b->stx - *stp);
    const double stpq = *stp + dp / (dp - b->dx) * (b->stx - *stp);
    stpf = fabs(stpc - *stp) > fabs(stpq - *stp) ? stpc : stpq;
    *brackt = 1;
  } else if (fabs(dp) < fabs(b->dx)) {
    /* lower value, same sign, derivative magnitude decreases */
    *info = 3;
    bound = 1;
    const double theta = 3 * (b->fx - fp) / (*stp - b->stx) + b->dx + dp;
    double gamma = mt_cubic_gamma(theta, b->dx, dp, 1);
    if (*stp > b->stx) gamma = -gamma;
    const double p = gamma - dp + theta;
    const double q = gamma + (b->dx - dp) + gamma;
    const double r = p / q;
    double stpc;
    if (r < 0 && gamma != 0) stpc = *stp + r * (b->stx - *stp);
    else stpc = *stp > b->stx ? stmax : stmin;
    const double stpq = *stp + dp / (dp - b->dx) * (b->stx - *stp);
    if (*brackt) stpf = fabs(*stp - stpc) < fabs(*stp - stpq) ? stpc : stpq;
    else stpf = fabs(*stp - stpc) > fabs(*stp - stpq) ? stpc : stpq;
  } else {
    /* lower value, same sign, derivative does not decrease */
    *info = 4;
    bound = 0;
    if (*brackt) {
      const double theta = 3 * (fp - b->fy) / (b->sty - *stp) + b->dy + dp;
      double gamma = mt_cubic_gamma(theta, b->dy, dp, 0);
      if (*stp > b->sty) gamma = -gamma;
      const double p = gamma - dp + theta;
      const double q = gamma - dp + gamma + b->dy;
      const double r = p / q;
      stpf = *stp + r * (b->sty - *stp);
    } else {
      stpf = *stp > b->stx ? stmax : stmin;
    }
  }
  /* bracket update */
  if (fp > b->fx) {
    b->sty = *stp; b->fy = fp; b->dy = dp;
  } else {
    if (sgnd < 0.0) { b->sty = b->stx; b->fy = b->fx; b->dy = b->dx; }
    b->stx = *stp; b->fx = fp; b->dx = dp;
  }
  /* safeguard */
  stpf = dmin(stmax, stpf);
  stpf = dmax(stmin, stpf);
  *stp = stpf;
  if (*brackt && bound) {
    if (b->sty > b->stx) *stp = dmin(b->stx + 0.66 * (b->sty - b->stx), *stp);
    else *stp = dmax(b->stx + 0.66 * (b->sty - b->stx), *stp);
  }
}

/* mcsrch (alglibinternal.cpp:12313-12632) with the evaluation inlined
 * instead of reverse communication.  Constants: alglibinternal.cpp:156-160.
 * Every evaluation is followed by trimfunction (optimization.cpp:9223-9241)
 * as mincgiteration does at optimization.cpp:17594. */
static void mt_search(long n, double* x, double* f, double* g, const double* d,
                      double* stp, double stpmax, double gtol, int* info,
                      int* nfev, double* wa, sro_fg_fn fg, void* ctx,
                      double trim_threshold) {
  const double ftol = 0.001, xtol = 100 * 2.220446049250313e-16 /* 100*eps */,
               stpmin = 1.0e-50, defstpmax = 1.0e+50, p5 = 0.5, p66 = 0.66,
               xtrapf = 4.0;
  const int maxfev = 20;
  /* ALGLIB's ae_machineepsilon is 5e-16 (ap.h), not DBL_EPSILON. */
  const double xtol_alglib = 100 * 5E-16;
  (void)xtol;
  if (stpmax == 0) stpmax = defstpmax;
  if (*stp < stpmin) *stp = stpmin;
  if (*stp > stpmax) *stp = stpmax;
  int infoc = 1;
  *info = 0;
  if (stpmax < stpmin && stpmax > 0) { *info = 5; *stp = stpmax; return; }
  if (n <= 0 || *stp <= 0 || stpmax < stpmin) return;
  const double dginit = vdot(g, d, n);
  if (dginit >= 0) return;
  int brackt = 0, stage1 = 1;
  *nfev = 0;
  const double finit = *f;
  const double dgtest = ftol * dginit;
  double width = stpmax - stpmin;
  double width1 = width / p5;
  memcpy(wa, x, sizeof(double) * n);
  mt_bracket b = {0, finit, dginit, 0, finit, dginit};
  double stmin = 0, stmax = 0;
  for (;;) {
    if (brackt) {
      if (b.stx < b.sty) { stmin = b.stx; stmax = b.sty; }
      else { stmin = b.sty; stmax = b.stx; }
    } else {
      stmin = b.stx;
      stmax = *stp + xtrapf * (*stp - b.stx);
    }
    if (*stp > stpmax) *stp = stpmax;
    if (*stp < stpmin) *stp = stpmin;
    if ((brackt && (*stp <= stmin || *stp >= stmax)) || *nfev >= maxfev - 1 ||
        infoc == 0 || (brackt && stmax - stmin <= xtol_alglib * stmax))
      *stp = b.stx;
    /* x = wa + stp*d  (ae_v_move + ae_v_addd) */
    for (long i = 0; i < n; ++i) { x[i] = wa[i]; x[i] += *stp * d[i]; }
    *f = fg(ctx, x, g);
    if (*f >= trim_threshold) { /* trimfunction */
      *f = trim_threshold;
      for (long i = 0; i < n; ++i) g[i] = 0.0;
    }
    *info = 0;
    *nfev += 1;
    const double dg = vdot(g, d, n);
    const double ftest1 = finit + *stp * dgtest;
    if ((brackt && (*stp <= stmin || *stp >= stmax)) || infoc == 0) *info = 6;
    if (*stp == stpmax && *f < finit && *f <= ftest1 && dg <= dgtest) *info = 5;
    if (*stp == stpmin && (*f >= finit || *f > ftest1 || dg >= dgtest)) *info = 4;
    if (*nfev >= maxfev) *info = 3;
    if (brackt && stmax - stmin <= xtol_alglib * stmax) *info = 2;
    if (*f < finit && *f <= ftest1 && fabs(dg) <= -gtol * dginit) *info = 1;
    if (*info != 0) {
      if (*info == 1 || *info == 5) {
        double v = 0.0;
        for (long i = 0; i < n; ++i) v += (wa[i] - x[i]) * (wa[i] - x[i]);
        if (*f >= finit || v == 0.0) *info = 6;
      }
      return;
    }
    if (stage1 && *f <= ftest1 && dg >= dmin(ftol, gtol) * dginit) stage1 = 0;
    if (stage1 && *f <= b.fx && *f > ftest1) {
      /* modified function psi(stp) = f(stp) - f(0) - ftol*stp*f'(0) */
      const double fm = *f - *stp * dgtest;
      mt_bracket m = {b.stx, b.fx - b.stx * dgtest, b.dx - dgtest,
                      b.sty, b.fy - b.sty * dgtest, b.dy - dgtest};
      const double dgm = dg - dgtest;
      mt_step(&m, stp, fm, dgm, &brackt, stmin, stmax, &infoc);
      b.stx = m.stx; b.sty = m.sty;
      b.fx = m.fx + m.stx * dgtest;
      b.fy = m.fy + m.sty * dgtest;
      b.dx = m.dx + dgtest;
      b.dy = m.dy + dgtest;
    } else {
      mt_step(&b, stp, *f, dg, &brackt, stmin, stmax, &infoc);
    }
    if (brackt) {
      if (fabs(b.sty - b.stx) >= p66 * width1) *stp = b.stx + p5 * (b.sty - b.stx);
      width1 = width;
      width = fabs(b.sty - b.stx);
    }
  }
}

void sro_mincg(int n_, double* x0, double epsg, double epsf, double epsx,
               int maxits, sro_fg_fn fg, sro_rep_fn rep, void* ctx,
               sro_cg_report* report) {
  const long n = n_;
  const double gtol = 0.3;        /* mincg_gtol, optimization.cpp:8904 */
  const int rscountdownlen = 10;  /* optimization.cpp:8903 */
  /* mincgsetcond: all-zero conditions select epsx = 1e-6
   * (optimization.cpp:16783-16786) */
  if (epsg == 0 && epsf == 0 && epsx == 0 && maxits == 0) epsx = 1.0E-6;
  double* x = (double*)xmalloc(sizeof(double) * n);
  double* g = (double*)xmalloc(sizeof(double) * n);
  double* xk = (double*)xmalloc(sizeof(double) * n);
  double* dk = (double*)xmalloc(sizeof(double) * n);
  double* xn = (double*)xmalloc(sizeof(double) * n);
  double* dn = (double*)xmalloc(sizeof(double) * n);
  double* d = (double*)xmalloc(sizeof(double) * n);
  double* yk = (double*)xmalloc(sizeof(double) * n);
  double* wa = (double*)xmalloc(sizeof(double) * n);
  memcpy(x, x0, sizeof(double) * n);
  int type = 0, its = 0, repnfev = 0;
  double f;
  /* optimization.cpp:17338-17349: first F/G at the start point */
  memcpy(xk, x, sizeof(double) * n);
  f = fg(ctx, x, g);
  const double trim = 10 * (fabs(f) + 1);  /* trimprepare */
  for (long i = 0; i < n; ++i) dk[i] = -g[i];
  if (rep) rep(ctx, x, f);  /* xupdated, stage 10 */
  double v = 0;
  for (long i = 0; i < n; ++i) v += (g[i] * 1.0) * (g[i] * 1.0);
  if (sqrt(v) <= epsg) {
    memcpy(xn, xk, sizeof(double) * n);
    type = 4;
    goto done;
  }
  repnfev = 1;
  double fold = f;
  double lastgoodstep = 1.0; /* no preconditioner, no suggested step */
  int rstimer = rscountdownlen;
  for (;;) {
    double stp;
    int mcinfo = 0, nfev = 0;
    for (long i = 0; i < n; ++i) yk[i] = -g[i];
    memcpy(d, dk, sizeof(double) * n);
    memcpy(x, xk, sizeof(double) * n);
    stp = 1.0;
    { /* linminnormalized, alglibinternal.cpp:12165-12196 */
      double mx = 0;
      for (long i = 0; i < n; ++i) mx = dmax(mx, fabs(d[i]));
      if (mx != 0) {
        double s = 1 / mx;
        for (long i = 0; i < n; ++i) d[i] *= s;
        stp = stp / s;
        s = vdot(d, d, n);
        s = 1 / sqrt(s);
        for (long i = 0; i < n; ++i) d[i] *= s;
        stp = stp / s;
      }
    }
    if (lastgoodstep != 0) stp = lastgoodstep;
    mt_search(n, x, &f, g, d, &stp, 0.0, gtol, &mcinfo, &nfev, wa, fg, ctx, trim);
    memcpy(xn, x, sizeof(double) * n);
    if (rep) rep(ctx, x, f);  /* xupdated, stage 19 */
    double betak;
    if (mcinfo == 1) {
      for (long i = 0; i < n; ++i) yk[i] += g[i];
      const double vv = vdot(yk, dk, n);
      const double betady = vdot(g, g, n) / vv;
      const double betahs = vdot(g, yk, n) / vv;
      betak = dmax(0.0, dmin(betady, betahs));  /* cgtype = 1 */
    } else {
      betak = 0;
    }
    if (its > 0 && its % (3 + n) == 0) betak = 0;
    if (mcinfo == 1 || mcinfo == 5) rstimer = rscountdownlen;
    else rstimer -= 1;
    for (long i = 0; i < n; ++i) dn[i] = -g[i];
    for (long i = 0; i < n; ++i) dn[i] += betak * dk[i];
    double lastscaledstep = 0.0;
    for (long i = 0; i < n; ++i) lastscaledstep += (d[i] / 1.0) * (d[i] / 1.0);
    lastscaledstep = stp * sqrt(lastscaledstep);
    if (mcinfo == 1) {
      lastgoodstep = 0;
      for (long i = 0; i < n; ++i) lastgoodstep += d[i] * d[i];
      lastgoodstep = stp * sqrt(lastgoodstep);
    }
    v = 0;
    for (long i = 0; i < n; ++i) v += (g[i] * 1.0) * (g[i] * 1.0);
    if (!isfinite(v) || !isfinite(f)) { type = -8; goto done; }
    repnfev += nfev;
    its += 1;
    if (its >= maxits && maxits > 0) { type = 5; goto done; }
    if (sqrt(v) <= epsg) { type = 4; goto done; }
    if (fold - f <= epsf * dmax(fabs(fold), dmax(fabs(f), 1.0))) { type = 1; goto done; }
    if (lastscaledstep <= epsx) { type = 2; goto done; }
    if (rstimer <= 0) { type = 7; goto done; }
    memcpy(xk, xn, sizeof(double) * n);
    memcpy(dk, dn, sizeof(double) * n);
    fold = f;
  }
done:
  memcpy(x0, xn, sizeof(double) * n);  /* mincgresults: X = XN */
  if (report) {
    report->termination_type = type;
    report->iterations = its;
    report->nfev = repnfev;
    report->f = f;
  }
  free(x); free(g); free(xk); free(dk); free(xn); free(dn); free(d); free(yk);
  free(wa);
}

/* ===================================================================== */
/* IRLS solver                                                           */
/* ===================================================================== */

void sro_irls_options_default(sro_irls_options* o) {
  /* map_solver.h:28-79, irls_map_solver.h:14-36 */
  o->max_num_solver_iterations = 50;
  o->gradient_norm_threshold = 1.0e-6;
  o->cost_decrease_threshold = 1.0e-6;
  o->parameter_variation_threshold = 1.0e-6;
  o->split_channels = 0;
  o->max_num_irls_iterations = 20;
  o->irls_cost_difference_threshold = 1.0e-5;
}

typedef struct {
  const sro_problem* p;
  int c0, c1;
  int nreg;
  double* weights[SRO_MAX_REG]; /* per round, (c1-c0)*N each */
} irls_ctx;

static double irls_objective(void* vctx, const double* x, double* g) {
  irls_ctx* c = (irls_ctx*)vctx;
  const long n = (long)c->p->W * c->p->H * (c->c1 - c->c0);
  for (long i = 0; i < n; ++i) g[i] = 0.0;
  double sum = data_term_range(c->p, c->c0, c->c1, x, g);
  for (int r = 0; r < c->nreg; ++r)
    sum += irls_term_range(c->p, r, c->c1 - c->c0, c->weights[r], x, g);
  return sum;
}

void sro_irls_solve(sro_problem* p, const sro_irls_options* opt,
                    const double* x0, double* x_out, sro_cg_fn cg,
                    sro_solve_report* report) {
  if (!cg) cg = sro_mincg;
  const long N = (long)p->W * p->H;
  const int C = p->C;
  /* irls_map_solver.cpp:200-216 */
  const int per_split = opt->split_channels ? 1 : C;
  const int rounds = C / per_split;
  const long npts = per_split * N;
  sro_irls_options o = *opt;
  double lambda_sum = 0.0;
  for (int r = 0; r < p->nreg; ++r) lambda_sum += p->lambda[r];
  { /* AdjustThresholdsAdaptively: int num_parameters * sum(lambda) */
    const double scale = (int)npts * lambda_sum;
    if (!(scale < 1.0)) {
      o.gradient_norm_threshold *= scale;
      o.cost_decrease_threshold *= scale;
      o.parameter_variation_threshold *= scale;
      o.irls_cost_difference_threshold *= scale;
    }
  }
  sro_solve_report rep = {0, 0, 0, 0.0};
  for (int round = 0; round < rounds; ++round) {
    const int c0 = round * per_split, c1 = c0 + per_split;
    double* x = (double*)xmalloc(sizeof(double) * npts);
    memcpy(x, x0 + c0 * N, sizeof(double) * npts);
    irls_ctx ctx;
    ctx.p = p; ctx.c0 = c0; ctx.c1 = c1; ctx.nreg = p->nreg;
    for (int r = 0; r < p->nreg; ++r) {
      ctx.weights[r] = (double*)xmalloc(sizeof(double) * npts);
      for (long i = 0; i < npts; ++i) ctx.weights[r][i] = 1.0;
    }
    /* RunIRLSLoop, irls_map_solver.cpp:45-157 */
    double previous_cost = INFINITY;
    double cost_difference = o.irls_cost_difference_threshold + 1.0;
    int ran = 0;
    while (fabs(cost_difference) >= o.irls_cost_difference_threshold) {
      sro_cg_report cr;
      cg((int)npts, x, o.gradient_norm_threshold, o.cost_decrease_threshold,
         o.parameter_variation_threshold, o.max_num_solver_iterations,
         irls_objective, NULL, &ctx, &cr);
      const double final_cost = cr.f;
      rep.cg_iterations += cr.iterations;
      rep.nfev += cr.nfev;
      rep.final_cost = final_cost;
      if (p->nreg == 0) { ran++; break; }
      for (int r = 0; r < p->nreg; ++r) {   /* :128-143 */
        double* vals = (double*)xmalloc(sizeof(double) * npts);
        sro_reg_values(&p->reg[r], x, p->W, p->H, per_split, vals);
        for (long i = 0; i < npts; ++i)
          ctx.weights[r][i] = 1.0 / dmax(0.00001, vals[i]);
        free(vals);
      }
      cost_difference = previous_cost - final_cost;
      previous_cost = final_cost;
      ran++;
      if (o.max_num_irls_iterations > 0 && ran >= o.max_num_irls_iterations)
        break;
    }
    rep.irls_rounds += ran;
    memcpy(x_out + c0 * N, x, sizeof(double) * npts);
    for (int r = 0; r < p->nreg; ++r) free(ctx.weights[r]);
    free(x);
  }
  if (report) *report = rep;
}

double sro_psnr(const double* gt, const double* im, long count) {
  double ssd = 0.0;
  for (long i = 0; i < count; ++i) {
    const double d = gt[i] - im[i];
    ssd += d * d;
  }
  const double mse = ssd / (double)count;
  return 20.0 * log10(1.0) - 10.0 * log10(mse);
}
