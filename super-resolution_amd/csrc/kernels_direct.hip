// kernels_direct.hip -- straightforward gfx950 kernels for every operator of
// the MAP gradient path: one thread per output element, inputs read through
// L1/L2.  They accept every geometry the reference accepts (arbitrary shifts,
// blur sizes, non-divisible HR sizes for ImageModel::ApplyToImage) and serve
//   (a) the operator entry points (srmap_apply, srmap_apply_transpose,
//       srmap_reg_values, srmap_reg_values_and_gradient), and
//   (b) the fallback of srmap_eval when the LDS-tiled kernels
//       (kernels_tiled.hip) do not cover the geometry, and their cross-check.
//
// Math: SURVEY.md section 8(a'); reference lines are cited per kernel.
#include <climits>
#include <cstdint>

#include "srmap_internal.hpp"

namespace srmap {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Sum over a 256-thread block; result valid in thread 0.
__device__ __forceinline__ double block_sum_256(double v, double* smem4) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) smem4[wid] = v;
  __syncthreads();
  double r = 0;
  if (threadIdx.x == 0) r = (smem4[0] + smem4[1]) + (smem4[2] + smem4[3]);
  return r;
}

template <typename T>
__device__ __forceinline__ WarpTaps<T> identity_warp() {
  WarpTaps<T> w;
  w.ox = 0; w.oy = 0; w.ntaps = 1; w.fx = 0; w.ytab = nullptr;
  w.w[0] = T(1); w.w[1] = T(0); w.w[2] = T(0); w.w[3] = T(0);
  return w;
}

// warped_k(rr, cc) for (rr, cc) already known to be inside the image:
// cv::warpAffine bilinear gather with zero border (motion_module.cpp:18-38).
template <typename T>
__device__ __forceinline__ T warp_sample(const T* __restrict__ plane, int W, int H,
                                         const WarpTaps<T>& wt, int rr, int cc) {
  int sr = rr + wt.oy;
  const int sc = cc + wt.ox;
  T w0 = wt.w[0], w1 = wt.w[1], w2 = wt.w[2], w3 = wt.w[3];
  if (wt.ytab != nullptr) {
    // per-row y table (rounding-tie shifts): BilinearTab_f's float32 products for this row's fraction index
    const int Y = wt.ytab[rr];
    sr = Y >> 5;
    const float tx1 = (float)wt.fx * (1.f / 32), tx0 = 1.f - tx1;
    const float ty1 = (float)(Y & 31) * (1.f / 32), ty0 = 1.f - ty1;
    w0 = (T)(ty0 * tx0); w1 = (T)(ty0 * tx1); w2 = (T)(ty1 * tx0); w3 = (T)(ty1 * tx1);
  } else if (wt.ntaps == 1) {
    return (sr >= 0 && sr < H && sc >= 0 && sc < W) ? plane[(size_t)sr * W + sc] : T(0);
  }
  const bool r0 = sr >= 0 && sr < H, r1 = sr + 1 >= 0 && sr + 1 < H;
  const bool c0 = sc >= 0 && sc < W, c1 = sc + 1 >= 0 && sc + 1 < W;
  const T v0 = (r0 && c0) ? plane[(size_t)sr * W + sc] : T(0);
  const T v1 = (r0 && c1) ? plane[(size_t)sr * W + sc + 1] : T(0);
  const T v2 = (r1 && c0) ? plane[(size_t)(sr + 1) * W + sc] : T(0);
  const T v3 = (r1 && c1) ? plane[(size_t)(sr + 1) * W + sc + 1] : T(0);
  return ((v0 * w0 + v1 * w1) + v2 * w2) + v3 * w3;
}

// ---------------------------------------------------------------------------
// Forward model A_k = D B M_k (image_model.cpp:86-91) at every LR pixel of
// frames [k0, k0+gridDim.z), optionally minus the observation, optionally with
// the data-term cost partial s^2 * sum(res^2) (objective_data_term.cpp:29-50).
template <typename T>
__global__ __launch_bounds__(256) void k_forward_direct(
    const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ out,
    double* __restrict__ partials, Geometry g,
    const WarpTaps<T>* __restrict__ warps, const T* __restrict__ blur,
    const int* __restrict__ col_map, const int* __restrict__ row_map, int k0,
    double cost_scale, int obs_C, int obs_c0) {
  __shared__ double red[4];
  __shared__ T comb[36];  // blur (x) bilinear taps of this block's frame, (b+1) x (b+1), b <= 5
  const int lp = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y, kk = blockIdx.z, k = k0 + kk;
  const int n = g.w * g.h;
  const WarpTaps<T> wt = warps ? warps[k] : identity_warp<T>();
  // interior fast path for bilinear warps: one (b+1)^2 stencil on x instead of b^2 four-tap samples
  const bool use_comb = wt.ntaps == 4 && wt.ytab == nullptr && g.b <= 5;
  const int nb1 = g.b + 1;
  if (use_comb) {
    if ((int)threadIdx.x < nb1 * nb1) {
      const int a1 = threadIdx.x / nb1, e1 = threadIdx.x - a1 * nb1;
      T s = T(0);
      for (int t = 0; t < 4; ++t) {
        const int a = a1 - (t >> 1), e = e1 - (t & 1);
        if (a >= 0 && a < g.b && e >= 0 && e < g.b) s += blur[a * g.b + e] * wt.w[t];
      }
      comb[threadIdx.x] = s;
    }
    __syncthreads();
  }
  double sq = 0.0;
  if (lp < n) {
    const int i = lp / g.w, j = lp - i * g.w;
    const int R0 = row_map[i], C0 = col_map[j];
    const T* plane = x + (size_t)c * g.W * g.H;
    T acc = T(0);
    const int sr0 = R0 - g.hb + wt.oy, sc0 = C0 - g.hb + wt.ox;
    if (use_comb && R0 - g.hb >= 0 && R0 + g.hb < g.H && C0 - g.hb >= 0 && C0 + g.hb < g.W && sr0 >= 0 &&
        sr0 + g.b < g.H && sc0 >= 0 && sc0 + g.b < g.W) {
      const T* src = plane + (size_t)sr0 * g.W + sc0;
      for (int a1 = 0; a1 < nb1; ++a1)
        for (int e1 = 0; e1 < nb1; ++e1) acc += comb[a1 * nb1 + e1] * src[(size_t)a1 * g.W + e1];
    } else {
      for (int a = 0; a < g.b; ++a) {
        const int rr = R0 + a - g.hb;
        if (rr < 0 || rr >= g.H) continue;  // filter2D BORDER_CONSTANT on the warped image
        for (int e = 0; e < g.b; ++e) {
          const int cc = C0 + e - g.hb;
          if (cc < 0 || cc >= g.W) continue;
          acc += blur[a * g.b + e] * warp_sample(plane, g.W, g.H, wt, rr, cc);
        }
      }
    }
    T res = acc;
    if (y) res -= y[((size_t)k * obs_C + c + obs_c0) * n + lp];
    out[((size_t)kk * g.C + c) * n + lp] = res;
    // cost rows (row-band sharding): LR row i counts when its first HR row is in [cr0, cr1)
    sq = (i * g.s >= g.cr0 && i * g.s < g.cr1) ? (double)res * (double)res : 0.0;
  }
  if (partials) {
    const double s = block_sum_256(sq, red);
    if (threadIdx.x == 0)
      partials[(size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] =
          cost_scale * s;
  }
}

template <typename T>
int launch_forward_direct(srmap_problem* p, const Geometry& g, const T* x, const T* y,
                          int obs_C, int obs_c0, T* out, int k0, int nk,
                          double* partials, int* nblocks, hipStream_t st) {
  dim3 grid((g.w * g.h + 255) / 256, g.C, nk);
  const double cost_scale = (double)g.s * (double)g.s;
  hipLaunchKernelGGL(k_forward_direct<T>, grid, dim3(256), 0, st, x, y, out, partials, g,
                     p->has_motion ? (const WarpTaps<T>*)p->d_fwd_warps : nullptr,
                     (const T*)p->d_blur, p->d_col_map, p->d_row_map, k0, cost_scale, obs_C,
                     obs_c0);
  if (nblocks) *nblocks = (int)(grid.x * grid.y * grid.z);
  SRMAP_HIP(p->ctx, hipGetLastError());
  return SRMAP_OK;
}

// ---------------------------------------------------------------------------
// Transpose model sum_k M_k^T B^T D^T r_k (image_model.cpp:93-101) in gather
// form at every HR pixel: zero-insertion upsample (image_data.cpp:99-115),
// correlation with kernel.t() (blur_module.cpp:30-36), warpAffine(-dx,-dy)
// (motion_module.cpp:40-51); each stage clipped to the H x W domain.
// SC: the scale at compile time (2, 3, 4; 0 = run time): the divisions / remainders by it in the tap loops are
// shifts and multiplies instead of ~40-instruction sequences (the ring mode was bound by them).
template <typename T, int SC>
__global__ __launch_bounds__(256) void k_gather_direct(
    const T* __restrict__ resid, T* __restrict__ gout, Geometry g,
    const WarpTaps<T>* __restrict__ warps, const T* __restrict__ blur_t, int k0,
    int nk, T out_scale, int accumulate, int ring, int ring_groups) {
  // ring mode: 256 / nfg pixels per block, the frames split over nfg thread groups (nfg = 16 for 16 frames and more:
  // one or a few frames per thread -- every frame costs two dependent memory round trips, its warp record and its
  // residuals, and with 4 groups a thread walked through 8 of them), combined through LDS in fixed order;
  // full mode: one pixel per thread, all frames
  const int gs = SC ? SC : g.s;
  __shared__ T part[256];
  const int nfg = ring > 0 ? ring_groups : 1, ppb = 256 / nfg;
  int hp = ring > 0 ? blockIdx.x * ppb + ((int)threadIdx.x % ppb) : blockIdx.x * 256 + threadIdx.x;
  const int fg = ring > 0 ? (int)threadIdx.x / ppb : 0;
  const int c = blockIdx.y;
  const int N = g.W * g.H, n = g.w * g.h;
  bool live = true;
  if (ring > 0) {
    // thread index -> pixel of the border ring of width `ring`: top band, bottom band, then the left / right strips
    const int band = ring * g.W, mid = g.H - 2 * ring, t = hp;
    live = t < 2 * band + 2 * ring * mid;
    int rr = 0, cc = 0;
    if (t < band) { rr = t / g.W; cc = t % g.W; }
    else if (t < 2 * band) { rr = g.H - ring + (t - band) / g.W; cc = (t - band) % g.W; }
    else if (live) { const int u = t - 2 * band, m = u % (2 * ring); rr = ring + u / (2 * ring); cc = m < ring ? m : g.W - 2 * ring + m; }
    hp = rr * g.W + cc;
  } else if (hp >= N) {
    return;
  }
  const int r = hp / g.W, col = hp - r * g.W;
  T acc = T(0);
  for (int kk = fg; kk < nk && live; kk += nfg) {
    const WarpTaps<T> wt = warps ? warps[k0 + kk] : identity_warp<T>();
    const T* rk = resid + ((size_t)kk * g.C + c) * n;
    T tk = T(0);
    int oy = wt.oy;
    T wloc[4] = {wt.w[0], wt.w[1], wt.w[2], wt.w[3]};
    if (wt.ytab != nullptr) {  // per-row y table of the transpose warp (rounding-tie shifts)
      const int Y = wt.ytab[r];
      oy = (Y >> 5) - r;
      const float tx1 = (float)wt.fx * (1.f / 32), tx0 = 1.f - tx1;
      const float ty1 = (float)(Y & 31) * (1.f / 32), ty0 = 1.f - ty1;
      wloc[0] = (T)(ty0 * tx0); wloc[1] = (T)(ty0 * tx1); wloc[2] = (T)(ty1 * tx0); wloc[3] = (T)(ty1 * tx1);
    }
    for (int t = 0; t < wt.ntaps; ++t) {
      const int pr = r + oy + (t >> 1), pc = col + wt.ox + (t & 1);
      if (pr < 0 || pr >= g.H || pc < 0 || pc >= g.W) continue;
      T v = T(0);
      // only the taps that land on the LR grid (every s-th), visited in the same increasing (a, e) order
      int a0 = (g.hb - pr) % gs, e0 = (g.hb - pc) % gs;
      if (a0 < 0) a0 += gs;
      if (e0 < 0) e0 += gs;
      for (int a = a0; a < g.b; a += gs) {
        const int R = pr + a - g.hb;
        if (R < 0 || R >= g.H) continue;
        const int li = R / gs;
        if (li >= g.h) continue;
        for (int e = e0; e < g.b; e += gs) {
          const int Cc = pc + e - g.hb;
          if (Cc < 0 || Cc >= g.W) continue;
          const int lj = Cc / gs;
          if (lj >= g.w) continue;
          v += blur_t[a * g.b + e] * rk[(size_t)li * g.w + lj];
        }
      }
      tk += wloc[t] * v;
    }
    acc += tk;
  }
  if (ring > 0) {
    part[threadIdx.x] = acc;
    __syncthreads();
    if (fg != 0 || !live) return;
    acc = part[threadIdx.x];
    for (int q = 1; q < nfg; ++q) acc += part[threadIdx.x + q * ppb];
  }
  const size_t o = (size_t)c * N + hp;
  const T base = accumulate ? gout[o] : T(0);
  gout[o] = base + out_scale * acc;
}

// The ring mode of the sub-pixel tile path as its own kernel: the pixels within `ring` of the image edge, the exact
// per-stage-clipped transpose (the expression of k_gather_direct), for blur sizes <= scale (at most one blur tap per
// dimension lands on the LR grid) and frames without per-row tables.  64 pixels x 4 frame groups per block: a WAVE is one
// frame group, so a frame's warp record comes through scalar loads; a thread owns up to FP frames and issues the requests
// of ALL of them (residual + blur coefficient per warp tap, at clamped addresses, the coefficient masked) before the first
// multiply-add -- in k_gather_direct every frame was two dependent round trips inside a branchy loop (12.8 us for 57 K
// pixels at cfg2 geometry).
template <typename T, int SC, int FP>
__global__ __launch_bounds__(256) void k_gather_ring(const T* __restrict__ resid, T* __restrict__ gout, Geometry g,
                                                    const WarpTaps<T>* __restrict__ warps, const T* __restrict__ blur_t,
                                                    int nk, T out_scale, int ring, T* __restrict__ ringbuf) {
  // ringbuf != nullptr: the ring's data gradient goes to ringbuf[c][t] (t = the ring pixel's number below) instead of
  // being added to gout -- the pass then runs AHEAD of the tile kernel, which adds the value as it stores g (and can
  // produce g.d with the gradient: kernels_ztile.hip, sub-pixel WD instances)
  constexpr int gs = SC;
  __shared__ T part[256];
  const int lane = threadIdx.x & 63;
  const int fg = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int c = blockIdx.y;
  const int N = g.W * g.H, n = g.w * g.h;
  // thread index -> pixel of the border ring of width `ring`: top band, bottom band, then the left / right strips
  const int band = ring * g.W, mid = g.H - 2 * ring, t = blockIdx.x * 64 + lane;
  const bool live = t < 2 * band + 2 * ring * mid;
  int rr = 0, cc = 0;
  if (t < band) { rr = t / g.W; cc = t % g.W; }
  else if (t < 2 * band) { rr = g.H - ring + (t - band) / g.W; cc = (t - band) % g.W; }
  else if (live) { const int u = t - 2 * band, m = u % (2 * ring); rr = ring + u / (2 * ring); cc = m < ring ? m : g.W - 2 * ring + m; }
  const int hp = rr * g.W + cc;
  typedef const WarpTaps<T> __attribute__((address_space(4))) * WP;
  const size_t o = (size_t)c * N + hp;
  const T gold = (fg == 0 && live && ringbuf == nullptr) ? gout[o] : T(0);  // requested with everything else (the launch adds to g)
  T rv[FP][4], cf[FP][4], wq[FP][4];
  // ---- every request of this thread's frames ----
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    const int kk = fg + 4 * f;
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) { rv[f][tp] = T(0); cf[f][tp] = T(0); wq[f][tp] = T(0); }
    if (kk >= nk) continue;  // uniform
    WP wt = (WP)(unsigned long long)(warps + kk);
    const int oy = wt->oy, ox = wt->ox, nt = wt->ntaps;
    const T* rk = resid + ((size_t)kk * g.C + c) * n;
    // row / column part of a tap's index, once per tap row / column: the warped pixel, the blur tap that lands on the LR
    // grid ((p + a - hb) a multiple of the scale; at most one since b <= scale), the LR index, and whether all of it exists
    int ia[2], ie[2], li[2], lj[2];
    bool okr[2], okc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int pr = rr + oy + d, pc = cc + ox + d;
      int a = (g.hb - pr) % gs, e = (g.hb - pc) % gs;
      if (a < 0) a += gs;
      if (e < 0) e += gs;
      const int R = pr + a - g.hb, Cc = pc + e - g.hb;
      ia[d] = a; ie[d] = e; li[d] = R / gs; lj[d] = Cc / gs;
      okr[d] = pr >= 0 && pr < g.H && a < g.b && R >= 0 && R < g.H && li[d] < g.h;
      okc[d] = pc >= 0 && pc < g.W && e < g.b && Cc >= 0 && Cc < g.W && lj[d] < g.w;
    }
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
      if (tp >= nt) continue;  // uniform
      wq[f][tp] = wt->w[tp];
      const int dy = tp >> 1, dx = tp & 1;
      const bool ok = live && okr[dy] && okc[dx];
      const T bt = blur_t[ok ? ia[dy] * g.b + ie[dx] : 0];
      rv[f][tp] = rk[ok ? li[dy] * g.w + lj[dx] : 0];
      cf[f][tp] = ok ? bt : T(0);
    }
  }
  // ---- the sums, frame by frame, tap by tap (k_gather_direct's order within a frame) ----
  T acc = T(0);
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    T tk = T(0);
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
      T v = T(0);
      v += cf[f][tp] * rv[f][tp];
      tk += wq[f][tp] * v;
    }
    acc += tk;
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (fg != 0 || !live) return;
  acc = part[lane];
#pragma unroll
  for (int q = 1; q < 4; ++q) acc += part[lane + q * 64];
  if (ringbuf != nullptr) ringbuf[(size_t)c * (size_t)(2 * band + 2 * ring * mid) + t] = out_scale * acc;
  else gout[o] = gold + out_scale * acc;
}

bool gather_ring_kernel_ok(const srmap_problem* p, const Geometry& geo, int nk, int ring) {
  return ring > 0 && 2 * ring < geo.H && 2 * ring < geo.W && p->has_motion && p->d_bwd_warps != nullptr && geo.b <= geo.s &&
         geo.s >= 2 && geo.s <= 4 && nk <= 16 && p->d_ytabs.empty();
}

template <typename T>
int launch_gather_direct(srmap_problem* p, const Geometry& geo, const T* resid, T* g,
                         int k0, int nk, double out_scale, bool accumulate,
                         hipStream_t st, int ring, T* ringbuf) {
  // ring > 0: only the pixels within `ring` of the image edge (the exact border of the sub-pixel tile path)
  size_t npix = (size_t)geo.W * geo.H;
  if (ring > 0) {
    if (2 * ring >= geo.H || 2 * ring >= geo.W) ring = 0;
    else npix = 2 * (size_t)ring * geo.W + 2 * (size_t)ring * (geo.H - 2 * ring);
  }
  const WarpTaps<T>* wp0 = p->has_motion ? (const WarpTaps<T>*)p->d_bwd_warps : nullptr;
  if (ringbuf != nullptr && !(accumulate && k0 == 0 && gather_ring_kernel_ok(p, geo, nk, ring)))
    return set_error(p->ctx, SRMAP_EINVAL, "internal: the ring buffer needs the ring kernel");
  if (ring > 0 && accumulate && k0 == 0 && gather_ring_kernel_ok(p, geo, nk, ring)) {
    dim3 grid((unsigned)((npix + 63) / 64), geo.C);
    const T* bt0 = (const T*)p->d_blur_t;
#define SRMAP_RING(SS, FF) hipLaunchKernelGGL((k_gather_ring<T, SS, FF>), grid, dim3(256), 0, st, resid, g, geo, wp0, bt0, nk, (T)out_scale, ring, ringbuf)
#define SRMAP_RING_S(SS) do { if (nk <= 4) SRMAP_RING(SS, 1); else if (nk <= 8) SRMAP_RING(SS, 2); else SRMAP_RING(SS, 4); } while (0)
    if (geo.s == 2) SRMAP_RING_S(2); else if (geo.s == 3) SRMAP_RING_S(3); else SRMAP_RING_S(4);
#undef SRMAP_RING_S
#undef SRMAP_RING
    SRMAP_HIP(p->ctx, hipGetLastError());
    return SRMAP_OK;
  }
  int groups = 1;
  while (groups < 16 && groups < nk) groups *= 2;  // thread groups of the ring mode: a power of two, at most 16
  const unsigned ppb = 256u / (unsigned)groups;
  dim3 grid((unsigned)(ring > 0 ? (npix + ppb - 1) / ppb : (npix + 255) / 256), geo.C);
  const WarpTaps<T>* wp = p->has_motion ? (const WarpTaps<T>*)p->d_bwd_warps : nullptr;
  const T* bt = (const T*)p->d_blur_t;
  const int acc1 = accumulate ? 1 : 0;
  if (geo.s == 2) hipLaunchKernelGGL((k_gather_direct<T, 2>), grid, dim3(256), 0, st, resid, g, geo, wp, bt, k0, nk, (T)out_scale, acc1, ring, groups);
  else if (geo.s == 3) hipLaunchKernelGGL((k_gather_direct<T, 3>), grid, dim3(256), 0, st, resid, g, geo, wp, bt, k0, nk, (T)out_scale, acc1, ring, groups);
  else if (geo.s == 4) hipLaunchKernelGGL((k_gather_direct<T, 4>), grid, dim3(256), 0, st, resid, g, geo, wp, bt, k0, nk, (T)out_scale, acc1, ring, groups);
  else hipLaunchKernelGGL((k_gather_direct<T, 0>), grid, dim3(256), 0, st, resid, g, geo, wp, bt, k0, nk, (T)out_scale, acc1, ring, groups);
  SRMAP_HIP(p->ctx, hipGetLastError());
  return SRMAP_OK;
}

// ---------------------------------------------------------------------------
// Regularizer values: TotalVariationRegularizer::ApplyToImage
// (tv_regularizer.cpp:110-132, helpers :21-106) and
// BilateralTotalVariationRegularizer::ApplyToImage (btv_regularizer.cpp:19-46,
// :67-90).
struct PowTable {
  double v[2 * kMaxBtvRange + 1];
};

template <typename T>
__device__ __forceinline__ T absval(T v) { return v < T(0) ? -v : v; }

template <typename T>
__device__ __forceinline__ T reg_value_at(const T* __restrict__ x, int W, int H, int C,
                                          int c, int r, int col, int kind, int range,
                                          const PowTable& pw, int zhi = 0) {
  const size_t N = (size_t)W * H;
  const T* plane = x + (size_t)c * N;
  const T x0 = plane[(size_t)r * W + col];
  if (kind == SRMAP_REG_BTV) {
    T tv = T(0);
    for (int i = 0; i <= range; ++i) {
      const int rr = r + i;
      if (rr >= H) continue;
      for (int j = 0; j <= range; ++j) {
        const int cc = col + j;
        if (cc >= W) continue;
        tv += (T)pw.v[i + j] * absval(x0 - plane[(size_t)rr * W + cc]);
      }
    }
    return tv;
  }
  const T yv = (r + 1 < H) ? absval(plane[(size_t)(r + 1) * W + col] - x0) : T(0);
  const T xv = (col + 1 < W) ? absval(plane[(size_t)r * W + col + 1] - x0) : T(0);
  T tv = yv + xv;
  if (kind == SRMAP_REG_TV3D && (c + 1 < C || zhi)) tv += absval(plane[N + (size_t)r * W + col] - x0);
  return tv;
}

template <typename T>
__global__ __launch_bounds__(256) void k_reg_values(const T* __restrict__ x,
                                                   T* __restrict__ values, int W, int H,
                                                   int C, int kind, int range,
                                                   PowTable pw, int zhi, int as_weights) {
  const int hp = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (hp >= W * H) return;
  const int r = hp / W, col = hp - r * W;
  T v = reg_value_at(x, W, H, C, c, r, col, kind, range, pw, zhi);
  if (as_weights) {  // w = 1 / max(1e-5, r) in the same pass (k_irls_weights' arithmetic)
    const T m = v > (T)0.00001 ? v : (T)0.00001;
    v = T(1) / m;
  }
  values[(size_t)c * W * H + hp] = v;
}

// BTV values (and IRLS weights) with FOUR consecutive pixels per thread: the (R + 1) x (R + 4) window of a thread comes as
// two 4-element vectors per row instead of (R + 1)^2 scalar loads per pixel (k_reg_values: 50 us per 2048^2 plane, 3 x per
// cfg2 solve).  Same taps in the same (i outer, j inner) order per pixel, skipped taps as exact zeros: bit-identical.
// Requires W % 4 == 0 and range R <= 3.
template <typename T, int R>
__global__ __launch_bounds__(256) void k_btv_values4(const T* __restrict__ x, T* __restrict__ values, int W, int H,
                                                    PowTable pw, int as_weights) {
  const int W4 = W >> 2;
  const int cell = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (cell >= W4 * H) return;
  const int r = cell / W4, c0 = (cell - r * W4) * 4;
  const T* plane = x + (size_t)c * W * H;
  const bool nin = c0 + 4 < W;  // the next cell of the row exists
  T tv[4] = {T(0), T(0), T(0), T(0)}, x0[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
  for (int i = 0; i <= R; ++i) {
    const int rr = r + i;
    const bool rin = rr < H;
    const T* row = plane + (size_t)(rin ? rr : r) * W + c0;
    T a[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = row[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[4 + q] = row[nin ? 4 + q : q];
    if (i == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) x0[q] = a[q];
    }
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
#pragma unroll
      for (int j = 0; j <= R; ++j) {
        const bool in = rin && (pc + j < 4 || nin);
        const T d = in ? x0[pc] - a[pc + j] : T(0);
        tv[pc] += (T)pw.v[i + j] * absval(d);
      }
    }
  }
  T* out = values + (size_t)c * W * H + (size_t)r * W + c0;
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) {
    T v = tv[pc];
    if (as_weights) {  // w = 1 / max(1e-5, r) (k_irls_weights' arithmetic)
      const T m = v > (T)0.00001 ? v : (T)0.00001;
      v = T(1) / m;
    }
    out[pc] = v;
  }
}

// The same with a thread walking DOWN a strip of RS rows: the window's R + 1 rows stay in registers and every output
// row requests ONE new row (two 4-element vectors) instead of R + 1 -- (RS + R) / RS rows read per row written instead
// of R + 1, a quarter of the memory instructions at R = 3.  Per pixel the same taps in the same order with the same
// zeros for the skipped ones as k_btv_values4: bit-identical (tests/test_gpu_parity.py, reg values against the CPU path).
#ifndef SRMAP_BTV_STRIP
#define SRMAP_BTV_STRIP 4
#endif
__device__ __forceinline__ float fabs_mod(float v) { return __builtin_fabsf(v); }
__device__ __forceinline__ double fabs_mod(double v) { return __builtin_fabs(v); }
// BORDER = false: the strip's windows stay inside the image (no per-tap masks: two f64 issues per tap instead of four).
template <typename T, int R, int RS, bool BORDER>
__device__ __forceinline__ void btv_strip_rows(const T* __restrict__ plane, T* __restrict__ oplane, int W, int H, int r0,
                                               int c0, bool nin, const PowTable& pw, int as_weights) {
  T win[R + 1][8];  // window rows r .. r + R of the row being written (rotating: row q lives in win[q % (R + 1)])
#pragma unroll
  for (int q = 0; q < RS + R; ++q) {
    // row r0 + q enters the window (rows below the image: a valid address, never used -- see `rin` below)
    {
      const int rr = r0 + q;
      const T* row = plane + (size_t)((!BORDER || rr < H) ? rr : r0) * W + c0;
#pragma unroll
      for (int e = 0; e < 4; ++e) win[q % (R + 1)][e] = row[e];
#pragma unroll
      for (int e = 0; e < 4; ++e) win[q % (R + 1)][4 + e] = row[(!BORDER || nin) ? 4 + e : e];
    }
    if (q < R) continue;
    const int o = q - R, r = r0 + o;  // the output row whose window is complete now
    if (BORDER && r >= H) continue;
    T tv[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int i = 0; i <= R; ++i) {
      const bool rin = r + i < H;
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) {
#pragma unroll
        for (int j = 0; j <= R; ++j) {
          T d = win[o % (R + 1)][pc] - win[(o + i) % (R + 1)][pc + j];
          if (BORDER) d = (rin && (pc + j < 4 || nin)) ? d : T(0);
          // |d| as the FMA's source modifier (absval's compare + select cost three more issues per tap).  The two differ
          // for d = -0 only, and a -0 product leaves the sum -- which starts at +0 -- unchanged just like a +0 one.
          tv[pc] += (T)pw.v[i + j] * fabs_mod(d);
        }
      }
    }
    T* out = oplane + (size_t)r * W + c0;
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
      T v = tv[pc];
      if (as_weights) {  // w = 1 / max(1e-5, r) (k_irls_weights' arithmetic)
        const T m = v > (T)0.00001 ? v : (T)0.00001;
        v = T(1) / m;
      }
      out[pc] = v;
    }
  }
}

template <typename T, int R, int RS>
__global__ __launch_bounds__(256) void k_btv_values_strip(const T* __restrict__ x, T* __restrict__ values, int W, int H,
                                                         PowTable pw, int as_weights) {
  const int W4 = W >> 2;
  const int nstrips = (H + RS - 1) / RS;
  const int cell = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (cell >= W4 * nstrips) return;
  const int sidx = cell / W4, c0 = (cell - sidx * W4) * 4;
  const int r0 = sidx * RS;
  const T* plane = x + (size_t)c * W * H;
  T* oplane = values + (size_t)c * W * H;
  const bool nin = c0 + 4 < W;  // the next cell of the row exists
  // one path per wave: only the waves that hold a row's last cell or the image's last strips take the masked one
  if (__all(nin && r0 + RS + R <= H)) btv_strip_rows<T, R, RS, false>(plane, oplane, W, H, r0, c0, nin, pw, as_weights);
  else btv_strip_rows<T, R, RS, true>(plane, oplane, W, H, r0, c0, nin, pw, as_weights);
}

template <typename T>
static bool launch_btv_values4(const Geometry& g, const RegSpec& rs, const T* x, T* out, int as_weights, const PowTable& pw,
                               hipStream_t st) {
  if (rs.kind != SRMAP_REG_BTV || rs.range < 1 || rs.range > 3 || (g.W & 3) != 0) return false;
  if ((long long)g.W * g.H / 4 >= (long long)INT_MAX) return false;
  if (SRMAP_BTV_STRIP > 0 && rs.range == 3 && g.H >= 4 * SRMAP_BTV_STRIP) {  // the solve's pass at cfg2
    constexpr int RS = SRMAP_BTV_STRIP > 0 ? SRMAP_BTV_STRIP : 1;
    const long long cells = (long long)(g.W >> 2) * ((g.H + RS - 1) / RS);
    dim3 sgrid((unsigned)((cells + 255) / 256), g.C);
    hipLaunchKernelGGL((k_btv_values_strip<T, 3, RS>), sgrid, dim3(256), 0, st, x, out, g.W, g.H, pw, as_weights);
    return true;
  }
  dim3 grid((unsigned)(((long long)(g.W >> 2) * g.H + 255) / 256), g.C);
  if (rs.range == 1) hipLaunchKernelGGL((k_btv_values4<T, 1>), grid, dim3(256), 0, st, x, out, g.W, g.H, pw, as_weights);
  else if (rs.range == 2) hipLaunchKernelGGL((k_btv_values4<T, 2>), grid, dim3(256), 0, st, x, out, g.W, g.H, pw, as_weights);
  else hipLaunchKernelGGL((k_btv_values4<T, 3>), grid, dim3(256), 0, st, x, out, g.W, g.H, pw, as_weights);
  return true;
}

static PowTable make_pow(const RegSpec& rs) {
  PowTable t;
  for (int i = 0; i < 2 * kMaxBtvRange + 1; ++i) t.v[i] = rs.pow_table[i];
  return t;
}

template <typename T>
int launch_reg_values(srmap_problem* p, const Geometry& g, const RegSpec& rs,
                      const T* x, T* values, hipStream_t st) {
  dim3 grid((g.W * g.H + 255) / 256, g.C);
  if (!launch_btv_values4<T>(g, rs, x, values, 0, make_pow(rs), st))
    hipLaunchKernelGGL(k_reg_values<T>, grid, dim3(256), 0, st, x, values, g.W, g.H, g.C,
                       rs.kind, rs.range, make_pow(rs), g.zhi, 0);
  SRMAP_HIP(p->ctx, hipGetLastError());
  return SRMAP_OK;
}

// The IRLS weights 1 / max(1e-5, regularizer(x)) of irls_map_solver.cpp:128-143 in one pass over x.
template <typename T>
int launch_reg_weights(srmap_problem* p, const Geometry& g, const RegSpec& rs,
                       const T* x, T* weights, hipStream_t st) {
  dim3 grid((g.W * g.H + 255) / 256, g.C);
  if (!launch_btv_values4<T>(g, rs, x, weights, 1, make_pow(rs), st))
    hipLaunchKernelGGL(k_reg_values<T>, grid, dim3(256), 0, st, x, weights, g.W, g.H, g.C,
                       rs.kind, rs.range, make_pow(rs), g.zhi, 1);
  SRMAP_HIP(p->ctx, hipGetLastError());
  return SRMAP_OK;
}

// ---------------------------------------------------------------------------
// Regularizer gradient given the values (tv_regularizer.cpp:143-224,
// btv_regularizer.cpp:105-166), bug-compatible: 3-D TV has no z self term; BTV
// uses the exclusive window and skips the absolute pixel (0,0) as a source.
// Constants c[q] = gc_scale * gc[q] (gc == nullptr means 1).
template <typename T>
__device__ __forceinline__ T sgn(T d) { return d > T(0) ? T(1) : (d < T(0) ? T(-1) : T(0)); }

template <typename T>
__global__ __launch_bounds__(256) void k_reg_gradient_direct(
    const T* __restrict__ x, const T* __restrict__ gc, T gc_scale,
    const T* __restrict__ values, T* __restrict__ gout, int accumulate,
    double* __restrict__ partials, int W, int H, int C, int kind, int range,
    PowTable pw, int cr0, int cr1) {
  __shared__ double red[4];
  const int hp = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  const size_t N = (size_t)W * H;
  double cost = 0.0;
  if (hp < W * H) {
    const int r = hp / W, col = hp - r * W;
    const T* plane = x + (size_t)c * N;
    const T* vals = values + (size_t)c * N;
    const T* gcp = gc ? gc + (size_t)c * N : nullptr;
    const size_t idx = (size_t)r * W + col;
    const T x0 = plane[idx];
    const T wt0 = gcp ? gcp[idx] : T(1);
    const T c0 = gc_scale * wt0;
    // values == nullptr (TV kinds only): the residual values are recomputed where they are needed -- the
    // k_reg_values pass and its C*N array round trip disappear (TV is 2-3 differences per value)
    const bool onfly = values == nullptr;
    const T r0 = onfly ? reg_value_at(x, W, H, C, c, r, col, kind, range, pw) : vals[idx];
    T grad = T(0);
    if (kind == SRMAP_REG_BTV) {
      T didi = T(0);
      for (int i = 0; i < range; ++i) {
        const int rr = r + i;
        if (rr >= H) continue;
        for (int j = 0; j < range; ++j) {
          const int cc = col + j;
          if (cc >= W) continue;
          didi += (T)pw.v[i + j] * sgn(x0 - plane[(size_t)rr * W + cc]);
        }
      }
      grad += T(2) * c0 * r0 * didi;
      for (int i = 0; i < range; ++i) {
        const int rr = r - i;
        if (rr < 0) continue;
        for (int j = 0; j < range; ++j) {
          const int cc = col - j;
          if (cc < 0 || (rr == 0 && cc == 0)) continue;
          const size_t q = (size_t)rr * W + cc;
          const T cq = gc_scale * (gcp ? gcp[q] : T(1));
          const T didj = -sgn(plane[q] - x0) * (T)pw.v[i + j];
          grad += T(2) * cq * vals[q] * didj;
        }
      }
    } else {
      T didi = T(0);
      if (col + 1 < W) didi -= sgn(plane[idx + 1] - x0);
      if (r + 1 < H) didi -= sgn(plane[idx + W] - x0);
      grad += T(2) * c0 * r0 * didi;
      if (col - 1 >= 0) {
        const size_t q = idx - 1;
        const T cq = gc_scale * (gcp ? gcp[q] : T(1));
        grad += T(2) * cq * (onfly ? reg_value_at(x, W, H, C, c, r, col - 1, kind, range, pw) : vals[q]) * sgn(x0 - plane[q]);
      }
      if (r - 1 >= 0) {
        const size_t q = idx - W;
        const T cq = gc_scale * (gcp ? gcp[q] : T(1));
        grad += T(2) * cq * (onfly ? reg_value_at(x, W, H, C, c, r - 1, col, kind, range, pw) : vals[q]) * sgn(x0 - plane[q]);
      }
      if (kind == SRMAP_REG_TV3D && c > 0) {
        const T xb = plane[idx - N];
        const T cq = gc_scale * (gcp ? gcp[idx - N] : T(1));
        grad += T(2) * cq * (onfly ? reg_value_at(x, W, H, C, c - 1, r, col, kind, range, pw) : vals[idx - N]) * sgn(x0 - xb);
      }
    }
    const size_t o = (size_t)c * N + idx;
    if (gout) gout[o] = (accumulate ? gout[o] : T(0)) + grad;
    // lambda * w * r^2  (objective_irls_regularization_term.cpp:45-50)
    cost = (r >= cr0 && r < cr1) ? (double)c0 * (double)r0 * (double)r0 : 0.0;
  }
  if (partials) {
    const double s = block_sum_256(cost, red);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
  }
}

// TV / 3-D TV in ONE pass (tv_regularizer.cpp:110-227): value, gradient and cost per pixel with the values of
// the left / upper / previous-channel neighbours recomputed on the spot (2-3 differences each) -- no values
// array, no integer division.  A thread owns 4 consecutive pixels of a row (block = 256 columns x 4 rows) and
// loads the row segments it needs once: rows r-1, r, r+1 of its plane, and for 3-D TV rows of the next and the
// previous channel.  Same expressions, in the same order, as reg_value_at + the TV branch of
// k_reg_gradient_direct.
template <typename T, bool D3>
__global__ __launch_bounds__(256) void k_tv_onepass(const T* __restrict__ x, const T* __restrict__ gc, T gc_scale,
                                                    T* __restrict__ gout, int accumulate,
                                                    double* __restrict__ partials, int W, int H, int C, int cr0,
                                                    int cr1, int zlo, int zhi) {
  __shared__ double red[4];
  constexpr int P = 4;
  const int c0 = (blockIdx.x * 64 + threadIdx.x) * P, r = blockIdx.y * 4 + threadIdx.y, c = blockIdx.z;
  const size_t N = (size_t)W * H;
  double cost = 0.0;
  if (c0 < W && r < H) {
    const T* plane = x + (size_t)c * N;
    const T* gcp = gc ? gc + (size_t)c * N : nullptr;
    // n values of row rr starting at column cs; positions outside the image read as 0 (never used unmasked)
    auto row = [&](const T* pl, int rr, int cs, int n, T* dst, T fill) {
      const bool rin = (unsigned)rr < (unsigned)H;
      const T* src = pl + (rin ? (size_t)rr * W : (size_t)0);
#pragma unroll
      for (int k = 0; k < P + 2; ++k)
        if (k < n) { const int cc = cs + k; dst[k] = (rin && (unsigned)cc < (unsigned)W) ? src[cc] : fill; }
    };
    T xc[P + 2], xd[P + 2], xu[P + 2], xn[P + 2], xnu[P + 2], xp[P + 2], xpd[P + 2], wc[P + 2], wu[P + 2], wp[P + 2];
    T gold[P];
#pragma unroll
    for (int i = 0; i < P; ++i) gold[i] = T(0);
    const bool has_next = D3 && (c + 1 < C || zhi), has_prev = D3 && (c > 0 || zlo);
    T* gdst = gout ? gout + (size_t)c * N + (size_t)r * W + c0 : nullptr;
    // rows r-1 .. r+1 inside the image and whole 4-pixel cells: every segment is one unconditional 4-vector plus
    // at most one clamped neighbour element, all requested before the first use (wave-uniform branch; the masked
    // element-wise form below made every load wait for the one before it)
    const bool fast = (W & 3) == 0 && r >= 1 && r + 1 < H;
    if (fast) {
      const int cl = c0 > 0 ? c0 - 1 : 0, cr = c0 + P < W ? c0 + P : W - 1;
      const size_t o0 = (size_t)r * W, od = o0 + W, ou = o0 - W;
      auto vec = [&](const T* src, T* dst) {
#pragma unroll
        for (int k = 0; k < P; ++k) dst[k] = src[k];
      };
      vec(plane + o0 + c0, xc + 1); xc[0] = plane[o0 + cl]; xc[P + 1] = plane[o0 + cr];
      vec(plane + od + c0, xd + 1); xd[0] = plane[od + cl];
      vec(plane + ou + c0, xu); xu[P] = plane[ou + cr];
      if (has_next) { vec(plane + N + o0 + c0, xn + 1); xn[0] = plane[N + o0 + cl]; vec(plane + N + ou + c0, xnu); }
      if (has_prev) { vec(plane - N + o0 + c0, xp); xp[P] = plane[o0 + cr - N]; vec(plane - N + od + c0, xpd); }
      if (gcp) {
        vec(gcp + o0 + c0, wc + 1); wc[0] = gcp[o0 + cl];
        vec(gcp + ou + c0, wu);
        if (has_prev) vec(gcp - N + o0 + c0, wp);
      }
      if (gdst && accumulate) vec(gdst, gold);
    } else {
      row(plane, r, c0 - 1, P + 2, xc, T(0));      // columns c0-1 .. c0+4
      row(plane, r + 1, c0 - 1, P + 1, xd, T(0));  // c0-1 .. c0+3
      row(plane, r - 1, c0, P + 1, xu, T(0));      // c0 .. c0+4
      if (has_next) { row(plane + N, r, c0 - 1, P + 1, xn, T(0)); row(plane + N, r - 1, c0, P, xnu, T(0)); }
      if (has_prev) { row(plane - N, r, c0, P + 1, xp, T(0)); row(plane - N, r + 1, c0, P, xpd, T(0)); }
      if (gcp) {
        row(gcp, r, c0 - 1, P + 1, wc, T(1));
        row(gcp, r - 1, c0, P, wu, T(1));
        if (has_prev) row(gcp - N, r, c0, P, wp, T(1));
      }
      if (gdst && accumulate) {
#pragma unroll
        for (int i = 0; i < P; ++i) if (c0 + i < W) gold[i] = gdst[i];
      }
    }
    const bool down = r + 1 < H;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int col = c0 + i;
      if (col < W) {
        const bool right = col + 1 < W;
        const T x0 = xc[i + 1];
        const T w0 = gcp ? wc[i + 1] : T(1);
        const T cc0 = gc_scale * w0;
        T r0;
        {
          const T yv = down ? absval(xd[i + 1] - x0) : T(0);
          const T xv = right ? absval(xc[i + 2] - x0) : T(0);
          r0 = yv + xv;
          if (has_next) r0 += absval(xn[i + 1] - x0);
        }
        T grad = T(0);
        T didi = T(0);
        if (right) didi -= sgn(xc[i + 2] - x0);
        if (down) didi -= sgn(xd[i + 1] - x0);
        grad += T(2) * cc0 * r0 * didi;  // 3-D TV has no z self term (tv_regularizer.cpp:154-170)
        if (col - 1 >= 0) {
          const T v0 = xc[i];
          const T yv = down ? absval(xd[i] - v0) : T(0);
          const T xv = absval(x0 - v0);  // col - 1 + 1 < W always
          T rq = yv + xv;
          if (has_next) rq += absval(xn[i] - v0);
          const T cq = gc_scale * (gcp ? wc[i] : T(1));
          grad += T(2) * cq * rq * sgn(x0 - v0);
        }
        if (r - 1 >= 0) {
          const T v0 = xu[i];
          const T yv = absval(x0 - v0);  // r - 1 + 1 < H always
          const T xv = right ? absval(xu[i + 1] - v0) : T(0);
          T rq = yv + xv;
          if (has_next) rq += absval(xnu[i] - v0);
          const T cq = gc_scale * (gcp ? wu[i] : T(1));
          grad += T(2) * cq * rq * sgn(x0 - v0);
        }
        if (has_prev) {
          const T v0 = xp[i];
          const T yv = down ? absval(xpd[i] - v0) : T(0);
          const T xv = right ? absval(xp[i + 1] - v0) : T(0);
          T rq = yv + xv;
          rq += absval(x0 - v0);  // (c - 1) + 1 < C always
          const T cq = gc_scale * (gcp ? wp[i] : T(1));
          grad += T(2) * cq * rq * sgn(x0 - v0);
        }
        if (gdst) gdst[i] = gold[i] + grad;
        // lambda * w * r^2  (objective_irls_regularization_term.cpp:45-50)
        cost += (r >= cr0 && r < cr1) ? (double)cc0 * (double)r0 * (double)r0 : 0.0;
      }
    }
  }
  if (partials) {  // (64, 4) block: wave = threadIdx.y
    const double ws = wave_sum(cost);
    if (threadIdx.x == 0) red[threadIdx.y] = ws;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0)
      partials[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// 3-D TV marching along the CHANNEL axis (tv_regularizer.cpp:110-227, the z terms :154-170 and :205-222).
// k_tv_onepass gives every (pixel, channel) its own thread, so each plane is fetched three times (as the current,
// the next and the previous channel) one whole plane of blocks apart -- from the Infinity Cache at best.  Here a
// thread keeps its 4 pixels and walks a chunk of channels with the planes c-1, c, c+1 (rows r-1, r, r+1 of each)
// in registers: per step it requests plane c+1, the weights and the old gradient of c -- all unconditional
// (out-of-range planes are clamped to a valid one and never used), one wait, then the arithmetic; other waves
// cover the wait.  Every plane, weight and gradient element crosses the fabric once per chunk.  The three plane
// sets rotate with compile-time indices (loop unrolled by 3).  Same expressions, same order as k_tv_onepass.
template <typename T> struct TvPlane { T m[6], d[5], u[5]; };  // row r: c0-1..c0+4, row r+1: c0-1..c0+3, row r-1: c0..c0+4

template <typename T>
__device__ __forceinline__ void tv_load4(const T* __restrict__ src, T* dst) {
  typedef T V4 __attribute__((ext_vector_type(4)));
  const V4 v = *reinterpret_cast<const V4*>(src);
  dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
}

template <typename T>
__device__ __forceinline__ void tv_load_plane(const T* __restrict__ pl, size_t o0, size_t od, size_t ou, int c0, int cl,
                                              int cr, TvPlane<T>& R) {
  tv_load4(pl + o0 + c0, R.m + 1); R.m[0] = pl[o0 + cl]; R.m[5] = pl[o0 + cr];
  tv_load4(pl + od + c0, R.d + 1); R.d[0] = pl[od + cl];
  tv_load4(pl + ou + c0, R.u); R.u[4] = pl[ou + cr];
}

template <typename T>
__global__ __launch_bounds__(256) void k_tv3d_march(const T* __restrict__ x, const T* __restrict__ gc, T gc_scale,
                                                    T* __restrict__ gout, int accumulate,
                                                    double* __restrict__ partials, int W, int H, int C, int cr0,
                                                    int cr1, int zlo, int zhi, int chunk) {
  __shared__ double red[4];
  constexpr int P = 4;
  const int c0 = (blockIdx.x * 64 + threadIdx.x) * P, r = blockIdx.y * 4 + threadIdx.y;
  const int cbeg = blockIdx.z * chunk, cend = (cbeg + chunk < C) ? cbeg + chunk : C;
  const ptrdiff_t N = (ptrdiff_t)W * H;
  double cost = 0.0;
  if (c0 < W && r < H) {  // W % 4 == 0: whole cells
    const bool up = r >= 1, down = r + 1 < H;
    const int cl = c0 >= 1 ? c0 - 1 : 0, cr = c0 + P < W ? c0 + P : W - 1;
    const size_t o0 = (size_t)r * W, od = o0 + (down ? W : 0), ou = o0 - (up ? W : 0);
    const bool cost_row = r >= cr0 && r < cr1;
    const int plo = zlo ? -1 : 0, phi = zhi ? C : C - 1;  // planes that exist
    auto clampc = [&](int c) { return c < plo ? plo : (c > phi ? phi : c); };
    TvPlane<T> R[3];
    T wprev[P];
    tv_load_plane(x + clampc(cbeg - 1) * N, o0, od, ou, c0, cl, cr, R[2]);
    tv_load_plane(x + cbeg * N, o0, od, ou, c0, cl, cr, R[0]);
#pragma unroll
    for (int i = 0; i < P; ++i) wprev[i] = T(1);
    if (gc) tv_load4(gc + clampc(cbeg - 1) * N + o0 + c0, wprev);
    for (int cb = cbeg; cb < cend; cb += 3) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = cb + k;
        if (c < cend) {  // uniform
          const TvPlane<T>& Rc = R[k];
          TvPlane<T>& Rn = R[(k + 1) % 3];
          const TvPlane<T>& Rp = R[(k + 2) % 3];
          tv_load_plane(x + clampc(c + 1) * N, o0, od, ou, c0, cl, cr, Rn);
          T wc[P + 1], wu[P], gold[P];
#pragma unroll
          for (int i = 0; i < P; ++i) { wc[i] = T(1); wu[i] = T(1); gold[i] = T(0); }
          wc[P] = T(1);
          if (gc) {
            const T* g0 = gc + c * N;
            tv_load4(g0 + o0 + c0, wc + 1); wc[0] = g0[o0 + cl];
            tv_load4(g0 + ou + c0, wu);
          }
          if (gout && accumulate) tv_load4(gout + c * N + o0 + c0, gold);
          const bool has_next = (c + 1 < C || zhi), has_prev = (c > 0 || zlo);
          T gnew[P];
#pragma unroll
          for (int i = 0; i < P; ++i) {
            const int col = c0 + i;
            const bool right = col + 1 < W;
            const T x0 = Rc.m[i + 1];
            const T cc0 = gc_scale * wc[i + 1];
            T r0;
            {
              const T yv = down ? absval(Rc.d[i + 1] - x0) : T(0);
              const T xv = right ? absval(Rc.m[i + 2] - x0) : T(0);
              r0 = yv + xv;
              if (has_next) r0 += absval(Rn.m[i + 1] - x0);
            }
            T grad = T(0);
            T didi = T(0);
            if (right) didi -= sgn(Rc.m[i + 2] - x0);
            if (down) didi -= sgn(Rc.d[i + 1] - x0);
            grad += T(2) * cc0 * r0 * didi;  // 3-D TV has no z self term (tv_regularizer.cpp:154-170)
            if (col - 1 >= 0) {
              const T v0 = Rc.m[i];
              const T yv = down ? absval(Rc.d[i] - v0) : T(0);
              const T xv = absval(x0 - v0);
              T rq = yv + xv;
              if (has_next) rq += absval(Rn.m[i] - v0);
              const T cq = gc_scale * wc[i];
              grad += T(2) * cq * rq * sgn(x0 - v0);
            }
            if (up) {
              const T v0 = Rc.u[i];
              const T yv = absval(x0 - v0);
              const T xv = right ? absval(Rc.u[i + 1] - v0) : T(0);
              T rq = yv + xv;
              if (has_next) rq += absval(Rn.u[i] - v0);
              const T cq = gc_scale * wu[i];
              grad += T(2) * cq * rq * sgn(x0 - v0);
            }
            if (has_prev) {
              const T v0 = Rp.m[i + 1];
              const T yv = down ? absval(Rp.d[i + 1] - v0) : T(0);
              const T xv = right ? absval(Rp.m[i + 2] - v0) : T(0);
              T rq = yv + xv;
              rq += absval(x0 - v0);
              const T cq = gc_scale * wprev[i];
              grad += T(2) * cq * rq * sgn(x0 - v0);
            }
            gnew[i] = gold[i] + grad;
            cost += cost_row ? (double)cc0 * (double)r0 * (double)r0 : 0.0;
          }
          if (gout) {
            typedef T V4 __attribute__((ext_vector_type(4)));
            V4 v; v.x = gnew[0]; v.y = gnew[1]; v.z = gnew[2]; v.w = gnew[3];
            *reinterpret_cast<V4*>(gout + c * N + o0 + c0) = v;
          }
#pragma unroll
          for (int i = 0; i < P; ++i) wprev[i] = wc[i + 1];
        }
      }
    }
  }
  if (partials) {  // (64, 4) block: wave = threadIdx.y
    const double ws = wave_sum(cost);
    if (threadIdx.x == 0) red[threadIdx.y] = ws;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0)
      partials[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

template <typename T>
int launch_reg_gradient_direct(srmap_problem* p, const Geometry& geo, const RegSpec& rs,
                               const T* x, const T* gc, double gc_scale, const T* values,
                               T* g, bool accumulate, double* partials, int* nblocks,
                               hipStream_t st) {
  if (values == nullptr && rs.kind != SRMAP_REG_BTV) {  // TV kinds, one pass
    dim3 grid2((geo.W + 255) / 256, (geo.H + 3) / 4, geo.C);
    const size_t valign = 4 * sizeof(T);
    if (rs.kind == SRMAP_REG_TV3D && geo.C >= 2 && (geo.W & 3) == 0 && ((uintptr_t)x % valign) == 0 &&
        (!gc || ((uintptr_t)gc % valign) == 0) && (!g || ((uintptr_t)g % valign) == 0)) {
      // channel march: chunks of channels per thread; enough chunks to fill the GPU at a few planes of overlap each
      const long long per_plane = (long long)grid2.x * grid2.y;
      int chunk = geo.C;
      while (chunk > 16 && per_plane * ((geo.C + chunk - 1) / chunk) < 4096) chunk = (chunk + 1) / 2;
      dim3 grid3(grid2.x, grid2.y, (geo.C + chunk - 1) / chunk);
      hipLaunchKernelGGL(k_tv3d_march<T>, grid3, dim3(64, 4), 0, st, x, gc, (T)gc_scale, g, accumulate ? 1 : 0, partials,
                         geo.W, geo.H, geo.C, geo.cr0, geo.cr1, geo.zlo, geo.zhi, chunk);
      if (nblocks) *nblocks = (int)(grid3.x * grid3.y * grid3.z);
      SRMAP_HIP(p->ctx, hipGetLastError());
      return SRMAP_OK;
    }
    if (rs.kind == SRMAP_REG_TV3D)
      hipLaunchKernelGGL((k_tv_onepass<T, true>), grid2, dim3(64, 4), 0, st, x, gc, (T)gc_scale, g, accumulate ? 1 : 0,
                         partials, geo.W, geo.H, geo.C, geo.cr0, geo.cr1, geo.zlo, geo.zhi);
    else
      hipLaunchKernelGGL((k_tv_onepass<T, false>), grid2, dim3(64, 4), 0, st, x, gc, (T)gc_scale, g, accumulate ? 1 : 0,
                         partials, geo.W, geo.H, geo.C, geo.cr0, geo.cr1, 0, 0);
    if (nblocks) *nblocks = (int)(grid2.x * grid2.y * grid2.z);
    SRMAP_HIP(p->ctx, hipGetLastError());
    return SRMAP_OK;
  }
  dim3 grid((geo.W * geo.H + 255) / 256, geo.C);
  hipLaunchKernelGGL(k_reg_gradient_direct<T>, grid, dim3(256), 0, st, x, gc, (T)gc_scale,
                     values, g, accumulate ? 1 : 0, partials, geo.W, geo.H, geo.C, rs.kind,
                     rs.range, make_pow(rs), geo.cr0, geo.cr1);
  if (nblocks) *nblocks = (int)(grid.x * grid.y);
  SRMAP_HIP(p->ctx, hipGetLastError());
  return SRMAP_OK;
}

// w = 1 / max(1e-5, r)  (irls_map_solver.cpp:128-143, kMinResidualValue :35)
template <typename T>
__global__ __launch_bounds__(256) void k_irls_weights(const T* __restrict__ values,
                                                     T* __restrict__ weights, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const T r = values[i];
  const T m = r > (T)0.00001 ? r : (T)0.00001;
  weights[i] = T(1) / m;
}

template <typename T>
int launch_irls_weights(srmap_problem* p, const T* values, T* weights, size_t n,
                        hipStream_t st) {
  hipLaunchKernelGGL(k_irls_weights<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                     values, weights, n);
  SRMAP_HIP(p->ctx, hipGetLastError());
  return SRMAP_OK;
}

// Deterministic final reduction of per-block partials, fixed order.  out[0] = sum.
// Up to kReduceChunk partials: one 256-thread block.  More (multi-channel images: one partial per tile and
// channel -- cfg5 has half a million): first one block per chunk of kReduceChunk partials into a scratch tail
// behind the partials, then one block over the chunk sums.  A single block over 73 K partials took 36 us.
constexpr int kReduceChunk = 4096;

__global__ __launch_bounds__(256) void k_reduce_partials(const double* __restrict__ partials,
                                                        int n, double* __restrict__ out) {
  __shared__ double red[4];
  const int i0 = blockIdx.x * kReduceChunk;
  const int i1 = (gridDim.x == 1) ? n : (i0 + kReduceChunk < n ? i0 + kReduceChunk : n);
  double v = 0.0;
  for (int i = i0 + threadIdx.x; i < i1; i += 256) v += partials[i];
  const double s = block_sum_256(v, red);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

int reduce_scratch_slots(size_t n) { return (int)((n + kReduceChunk - 1) / kReduceChunk); }

// `partials` must have room for n + reduce_scratch_slots(n) doubles.
int launch_reduce_partials(srmap_problem* p, const double* partials, int n, double* out,
                           hipStream_t st) {
  if (n <= kReduceChunk) {
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, st, partials, n, out);
  } else {
    const int nb = reduce_scratch_slots((size_t)n);
    double* scratch = const_cast<double*>(partials) + n;
    hipLaunchKernelGGL(k_reduce_partials, dim3(nb), dim3(256), 0, st, partials, n, scratch);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, st, (const double*)scratch, nb, out);
  }
  SRMAP_HIP(p->ctx, hipGetLastError());
  return SRMAP_OK;
}

#define INSTANTIATE(T)                                                                      \
  template int launch_forward_direct<T>(srmap_problem*, const Geometry&, const T*,         \
                                        const T*, int, int, T*, int, int, double*, int*,   \
                                        hipStream_t);                                       \
  template int launch_gather_direct<T>(srmap_problem*, const Geometry&, const T*, T*, int, \
                                       int, double, bool, hipStream_t, int, T*);            \
  template int launch_reg_values<T>(srmap_problem*, const Geometry&, const RegSpec&,       \
                                    const T*, T*, hipStream_t);                             \
  template int launch_reg_gradient_direct<T>(srmap_problem*, const Geometry&,              \
                                             const RegSpec&, const T*, const T*, double,   \
                                             const T*, T*, bool, double*, int*,            \
                                             hipStream_t);                                  \
  template int launch_irls_weights<T>(srmap_problem*, const T*, T*, size_t, hipStream_t);       \
  template int launch_reg_weights<T>(srmap_problem*, const Geometry&, const RegSpec&, const T*, T*, hipStream_t);
INSTANTIATE(float)
INSTANTIATE(double)

}  // namespace srmap
