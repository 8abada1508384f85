// registration.hip -- translational registration of a frame stack on the GPU.
//
// Reference interface: registration::TranslationalRegistration (src/motion/registration.cpp:161-201): shift
// (dx, dy) of every image relative to the first, image_i(p + d_i) = image_0(p); the first image gets (0, 0).  The
// reference gets there through OpenCV feature matching (BRISK + FLANN + RANSAC homography +
// estimateRigidTransform, :41-157), none of which is part of this path or available here; its own test
// (test/test_registration.cpp) fixes the CONTRACT: shifts applied with MotionModule are recovered to 0.01 px.
// This is a dense, GPU-native estimator for the same contract:
//   1. box pyramid of every frame (k_down2) down to <= 64 px (shifts up to a quarter of the frame are searched);
//   2. integer shift, coarse to fine: mean squared difference over the overlap for the (2R+1)^2 candidates around
//      twice the coarser level's answer (k_ssd_candidates: one workgroup per candidate x row chunk, fixed-order
//      reduction) -- the zero border warpAffine leaves in a shifted frame is inside the overlap of the TRUE shift
//      only where both frames agree, so the true integer shift has error exactly 0;
//   3. sub-pixel refinement: Gauss-Newton on sum (I_i(p + d) - I_0(p))^2 with the template's central-difference
//      gradients (k_lk_sums: bilinear sample, 5 sums per workgroup), pixels within |d| + 2 of the border left out.
// All sums are reduced in index order on the host from per-workgroup partials (deterministic).
#include <algorithm>
#include <cmath>
#include <vector>

#include "srmap_internal.hpp"

namespace srmap {

namespace {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// dst[h2][w2] = mean of the 2 x 2 blocks of src[h][w] (w2 = w / 2, h2 = h / 2)
__global__ __launch_bounds__(256) void k_down2(const double* __restrict__ src, double* __restrict__ dst, int w, int w2,
                                               int h2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= w2 * h2) return;
  const int r = i / w2, c = i - r * w2;
  const double* s = src + (size_t)(2 * r) * w + 2 * c;
  dst[i] = 0.25 * ((s[0] + s[1]) + (s[w] + s[w + 1]));
}

// partial[(cand * gridDim.y + chunk) * 2 + {0, 1}] = sum of (b(p + u) - a(p))^2 and the pixel count over the rows of
// this chunk, u = (ux0 + cand % n1, uy0 + cand / n1).
__global__ __launch_bounds__(256) void k_ssd_candidates(const double* __restrict__ a, const double* __restrict__ b, int w,
                                                        int h, int ux0, int uy0, int n1, int rows_per_chunk,
                                                        double* __restrict__ partial) {
  __shared__ double red[2][4];
  const int cand = blockIdx.x, ux = ux0 + cand % n1, uy = uy0 + cand / n1;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(h, r0 + rows_per_chunk);
  const int c_lo = max(0, -ux), c_hi = min(w, w - ux);  // columns with p + u inside
  double s = 0.0, n = 0.0;
  for (int r = r0; r < r1; ++r) {
    const int rb = r + uy;
    if (rb < 0 || rb >= h) continue;  // uniform
    for (int c = c_lo + (int)threadIdx.x; c < c_hi; c += 256) {
      const double d = b[(size_t)rb * w + c + ux] - a[(size_t)r * w + c];
      s += d * d;
      n += 1.0;
    }
  }
  s = wsum(s); n = wsum(n);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[0][wv] = s; red[1][wv] = n; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = partial + ((size_t)cand * gridDim.y + blockIdx.y) * 2;
    o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// Gauss-Newton sums of the translation-only Lucas-Kanade step at shift (dx, dy):
//   e = b(p + d) - a(p) (bilinear), gx, gy = central differences of a;  partial[block][6] = sum gx^2, gx gy, gy^2,
//   gx e, gy e, e^2 over the pixels at least `margin` from every edge.
__global__ __launch_bounds__(256) void k_lk_sums(const double* __restrict__ a, const double* __restrict__ b, int w, int h,
                                                 double dx, double dy, int margin, double* __restrict__ partial) {
  __shared__ double red[6][4];
  const int r = margin + blockIdx.x;  // one row per workgroup
  double acc[6] = {0, 0, 0, 0, 0, 0};
  const double fx = floor(dx), fy = floor(dy);
  const int ix = (int)fx, iy = (int)fy;
  const double tx = dx - fx, ty = dy - fy;
  if (r < h - margin) {
    for (int c = margin + (int)threadIdx.x; c < w - margin; c += 256) {
      const double* pa = a + (size_t)r * w + c;
      const double gx = 0.5 * (pa[1] - pa[-1]), gy = 0.5 * (pa[w] - pa[-w]);
      const double* pb = b + (size_t)(r + iy) * w + c + ix;
      const double v = (1.0 - ty) * ((1.0 - tx) * pb[0] + tx * pb[1]) + ty * ((1.0 - tx) * pb[w] + tx * pb[w + 1]);
      const double e = v - pa[0];
      acc[0] += gx * gx; acc[1] += gx * gy; acc[2] += gy * gy; acc[3] += gx * e; acc[4] += gy * e; acc[5] += e * e;
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const double s = wsum(acc[q]);
    if (lane == 0) red[q][wv] = s;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int q = threadIdx.x;
    partial[(size_t)blockIdx.x * 6 + q] = (red[q][0] + red[q][1]) + (red[q][2] + red[q][3]);
  }
}

}  // namespace

}  // namespace srmap

using namespace srmap;

extern "C" int srmap_register_translational(srmap_ctx* ctx, int num_images, int width, int height,
                                            const double* images_host, double* shifts_xy_out) {
  return srmap_register_translational_ex(ctx, num_images, width, height, images_host, shifts_xy_out, nullptr);
}

// quality_out (optional): 2 doubles per image --
//   [2i]     separation = 1 - best / runner-up of the coarsest search, the runner-up being the smallest mean squared
//            difference among candidates at least 2 coarse pixels away from the best one: near 1 = one clear minimum,
//            near 0 = ambiguous (periodic texture, no texture, motion that is not a translation);
//   [2i + 1] root mean squared residual I_i(p + d) - I_0(p) at the returned shift (same units as the pixels).
extern "C" int srmap_register_translational_ex(srmap_ctx* ctx, int num_images, int width, int height,
                                               const double* images_host, double* shifts_xy_out, double* quality_out) {
  if (!ctx || !shifts_xy_out || num_images < 0) return SRMAP_EINVAL;
  if (num_images == 0) return SRMAP_OK;  // registration.cpp:165-168: empty sequence
  if (!images_host || width < 8 || height < 8) return set_error(ctx, SRMAP_EINVAL, "registration needs images of at least 8 x 8");
  SRMAP_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t npx = (size_t)width * height;
  shifts_xy_out[0] = 0.0; shifts_xy_out[1] = 0.0;  // registration.cpp:170-172
  if (quality_out) { quality_out[0] = 1.0; quality_out[1] = 0.0; }
  if (num_images == 1) return SRMAP_OK;

  // level sizes
  std::vector<int> lw{width}, lh{height};
  while (std::min(lw.back(), lh.back()) > 64 && lw.size() < 12) { lw.push_back(lw.back() / 2); lh.push_back(lh.back() / 2); }
  const int L = (int)lw.size();
  size_t pyr_elems = 0;
  for (int l = 0; l < L; ++l) pyr_elems += (size_t)lw[l] * lh[l];
  double *d_a = nullptr, *d_b = nullptr, *d_part = nullptr;
  // coarsest level: shifts up to a quarter of the frame (16 coarse pixels at most); finer levels: +-1
  const int R0 = std::max(4, std::min(16, std::min(lw.back(), lh.back()) / 4)), n1c = 2 * R0 + 1;
  const int max_chunks = 64;
  const size_t part_elems = std::max((size_t)n1c * n1c * max_chunks * 2, (size_t)height * 6);
  std::vector<double> h_part(part_elems);
  int rc = SRMAP_OK;
  auto fail = [&](int code, const char* what) {
    rc = set_error(ctx, code, "registration: %s", what);
  };
  if (hipMalloc((void**)&d_a, pyr_elems * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&d_b, pyr_elems * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&d_part, part_elems * sizeof(double)) != hipSuccess) {
    fail(SRMAP_ENOMEM, "device allocation failed");
  }
  std::vector<size_t> off(L, 0);
  for (int l = 1; l < L; ++l) off[l] = off[l - 1] + (size_t)lw[l - 1] * lh[l - 1];
  auto build = [&](double* base, const double* host) -> bool {
    if (hipMemcpyAsync(base, host, npx * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) return false;
    for (int l = 1; l < L; ++l) {
      const int n = lw[l] * lh[l];
      hipLaunchKernelGGL(k_down2, dim3((n + 255) / 256), dim3(256), 0, st, base + off[l - 1], base + off[l], lw[l - 1], lw[l], lh[l]);
    }
    return hipGetLastError() == hipSuccess;
  };
  if (rc == SRMAP_OK && !build(d_a, images_host)) fail(SRMAP_EHIP, "pyramid of the reference frame failed");

  for (int i = 1; i < num_images && rc == SRMAP_OK; ++i) {
    if (!build(d_b, images_host + (size_t)i * npx)) { fail(SRMAP_EHIP, "pyramid failed"); break; }
    // ---- integer shift, coarse to fine ----
    int sx = 0, sy = 0;
    for (int l = L - 1; l >= 0 && rc == SRMAP_OK; --l) {
      const int R = (l == L - 1) ? R0 : 1, n1 = 2 * R + 1, ncand = n1 * n1;
      if (l != L - 1) { sx *= 2; sy *= 2; }
      const int chunks = std::min(max_chunks, std::max(1, lh[l] / 16));
      const int rpc = (lh[l] + chunks - 1) / chunks;
      hipLaunchKernelGGL(k_ssd_candidates, dim3(ncand, chunks), dim3(256), 0, st, d_a + off[l], d_b + off[l], lw[l], lh[l],
                         sx - R, sy - R, n1, rpc, d_part);
      if (hipMemcpyAsync(h_part.data(), d_part, (size_t)ncand * chunks * 2 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess) { fail(SRMAP_EHIP, "candidate search failed"); break; }
      double best = 0.0; int bi = -1;
      std::vector<double> msd(ncand, -1.0);
      for (int cnd = 0; cnd < ncand; ++cnd) {
        double s = 0.0, n = 0.0;
        for (int k = 0; k < chunks; ++k) { s += h_part[((size_t)cnd * chunks + k) * 2]; n += h_part[((size_t)cnd * chunks + k) * 2 + 1]; }
        if (n < 0.25 * lw[l] * lh[l]) continue;  // overlap too small to mean anything
        const double m = s / n;
        msd[cnd] = m;
        if (bi < 0 || m < best) { best = m; bi = cnd; }
      }
      if (bi < 0) { fail(SRMAP_EINVAL, "Could not determine motion shift between images."); break; }  // registration.cpp:193-194
      if (l == L - 1 && quality_out) {
        double runner = -1.0;
        for (int cnd = 0; cnd < ncand; ++cnd) {
          if (msd[cnd] < 0 || std::max(std::abs(cnd % n1 - bi % n1), std::abs(cnd / n1 - bi / n1)) < 2) continue;
          if (runner < 0 || msd[cnd] < runner) runner = msd[cnd];
        }
        quality_out[2 * i] = runner > 0 ? 1.0 - best / runner : 0.0;
      }
      sx = sx - R + bi % n1;
      sy = sy - R + bi / n1;
    }
    if (rc != SRMAP_OK) break;
    // ---- sub-pixel refinement at full resolution ----
    double dx = sx, dy = sy, rms = -1.0;
    for (int it = 0; it < 20; ++it) {
      const int margin = (int)std::ceil(std::max(std::fabs(dx), std::fabs(dy))) + 2;
      const int rows = height - 2 * margin;
      if (rows < 4 || width - 2 * margin < 4) break;  // keep the integer estimate
      hipLaunchKernelGGL(k_lk_sums, dim3(rows), dim3(256), 0, st, d_a, d_b, width, height, dx, dy, margin, d_part);
      if (hipMemcpyAsync(h_part.data(), d_part, (size_t)rows * 6 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess) { fail(SRMAP_EHIP, "refinement failed"); break; }
      double S[6] = {0, 0, 0, 0, 0, 0};
      for (int r = 0; r < rows; ++r)
        for (int q = 0; q < 6; ++q) S[q] += h_part[(size_t)r * 6 + q];
      rms = std::sqrt(S[5] / ((double)rows * (width - 2 * margin)));  // at the (dx, dy) this pass evaluated
      const double det = S[0] * S[2] - S[1] * S[1];
      if (!(det > 1e-12 * (S[0] * S[2] + 1e-300))) break;  // no texture: integer estimate stands
      const double ux = -(S[2] * S[3] - S[1] * S[4]) / det, uy = -(S[0] * S[4] - S[1] * S[3]) / det;
      // stay within the pixel the search found
      dx = std::min(std::max(dx + ux, sx - 1.0), sx + 1.0);
      dy = std::min(std::max(dy + uy, sy - 1.0), sy + 1.0);
      if (std::fabs(ux) < 1e-5 && std::fabs(uy) < 1e-5) break;
    }
    shifts_xy_out[2 * i] = dx;
    shifts_xy_out[2 * i + 1] = dy;
    if (quality_out) quality_out[2 * i + 1] = rms;
  }
  if (d_a) (void)hipFree(d_a);
  if (d_b) (void)hipFree(d_b);
  if (d_part) (void)hipFree(d_part);
  return rc;
}
