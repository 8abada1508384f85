// tiled_device.hpp -- pieces shared by the tiled evaluation kernels
// (kernels_tiled.hip): tile geometry, the polyphase-LDS forward stencil,
// the gather of one frame, the regulariser passes, and the host-side tile plan.
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "srmap_internal.hpp"

namespace srmap {

// Per-frame shift decomposition, precomputed on the host.
struct FrameInfo {
  int frow;   // forward: tile-row offset  S*i0 + oy + hu - hb   (x rows)
  int fcell;  // forward: cell offset      j0 + hlc + floor(ox / S)
  int fxm;    // forward: ox mod S  (0..S-1)
  int sy, sx; // gather: residual (li, lj) of the LR region is stored at (li - sy, lj - sx) so that every
              // frame's 2 x 2 patch of cell (ci, cj) sits at rows ci, ci+1 / columns cj, cj+1
  int gym;    // gather: toy mod S
  int gxm;    // gather: tox mod S
  int toy, tox;  // transpose integer offsets (for the border test)
};

// integer floor division / modulo on the host
inline int fdiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
inline int pmod(int a, int b) { return a - fdiv(a, b) * b; }

struct HostPlan {
  bool ok = false;
  int hu = 0, hd = 0, hlc = 0, hrc = 0, i0 = 0, j0 = 0, lrh = 0, lrw = 0, margin = 0;
  int regk = 0, regr = 0, reg_index = -1;
  std::vector<FrameInfo> frames;
};

struct PlanCache {
  HostPlan plan;
  FrameInfo* d_frames = nullptr;
  int* d_gb = nullptr;   // [K][S]
  void* d_wr = nullptr;  // [K][S][2] dtype
  void* d_wc = nullptr;  // [K][S][2] dtype
};

// plan of a problem (built by tiled_plan), nullptr if the tiled kernels do not cover it
PlanCache* tiled_find_plan(const srmap_problem* p);

namespace {


constexpr int kMaxHaloRows = 12;  // max hu + hd of the x tile
constexpr int kMaxHaloCells = 4;  // max hlc + hrc
constexpr int kTabFrames = 64;         // frames whose gather weights are staged in LDS (cfg5 has 64)
constexpr unsigned kSubCounters = 32;  // first-level arrival counters of the in-kernel cost reduction

constexpr int cmax_(int a, int b) { return a > b ? a : b; }

template <typename T, int S>
struct TileCfg {
  static constexpr int CW = 64;                        // LR cells per tile row = lanes
  static constexpr int TH = (S == 3) ? 9 : 8;          // HR rows per tile = waves
  static constexpr int NW = TH;
  static constexpr int NT = 64 * NW;                   // threads per workgroup
  static constexpr int CH = TH / S;                    // LR cell rows per tile
  static constexpr int TW = CW * S;
  static constexpr int XR = TH + kMaxHaloRows;         // x rows held in LDS
  static constexpr int XCELLS = CW + kMaxHaloCells;    // cells per x row
  static constexpr int XPLANE = XCELLS;                // elements per (row, phase)
  static constexpr int XROW = S * XPLANE;              // elements per row
  static constexpr int LRH = CH + 3, LRW = CW + 3;     // LR residual region (max)
  static constexpr int FR = NW;                        // frames per round (one frame per wave)
  static constexpr int MAXJ = LRH + 1;                 // sweeps per wave: one per LR row (lanes = first 64
                                                       // LR columns) + one tail sweep for columns 64..LRW-1
  static constexpr int CRPLANE = CW + 1;               // 2*lambda*w*r: one halo cell column
  static constexpr int CRROW = S * CRPLANE;
  static constexpr int XS_ELEMS = XR * XROW;
  static constexpr int GRH = CH + 2, GRW = CW + 2;     // residuals as the gather reads them (frame-aligned)
  static constexpr int RS_ELEMS = FR * GRH * GRW;
};



__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

constexpr int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
constexpr int posmod(int a, int b) { return a - floordiv(a, b) * b; }

// sgn(d) * pw with sgn(0) = 0 (pw > 0).  f32: ldexp pushes every non-zero d
// (subnormals included) beyond pw, med3 clamps to +-pw: 2 VALU ops, exact
// (pw <= 2^50 assumed: BTV decay powers and 1).
template <typename T>
__device__ __forceinline__ T sgn_scaled(T d, T pw) { return d > T(0) ? pw : (d < T(0) ? -pw : T(0)); }
template <>
__device__ __forceinline__ float sgn_scaled<float>(float d, float pw) {
  return __builtin_amdgcn_fmed3f(__builtin_ldexpf(d, 200), -pw, pw);
}
// f64: same idea with ldexp + min + max (3 full-rate ops; the compare/select form costs 2 v_cmp_f64 +
// 4 v_cndmask_b32, and compare->select pairs stall the issue port, tools/ubench/valu_asm.hip).
template <>
__device__ __forceinline__ double sgn_scaled<double>(double d, double pw) {
  return __builtin_fmin(__builtin_fmax(__builtin_ldexp(d, 1200), -pw), pw);
}
template <typename T>
__device__ __forceinline__ T sgnv(T d) { return sgn_scaled<T>(d, T(1)); }
template <typename T>
__device__ __forceinline__ T absv(T d) { return d < T(0) ? -d : d; }
template <>
__device__ __forceinline__ float absv<float>(float d) { return __builtin_fabsf(d); }
template <>
__device__ __forceinline__ double absv<double>(double d) { return __builtin_fabs(d); }

// ---- forward residual of ONE LR pixel for a frame whose ox mod S == OXM ----
template <typename T, int S, int B, int OXM, bool BORDER>
__device__ __forceinline__ T forward_taps(const T* __restrict__ xs, int addr,
                                          const T (&blur)[B * B], unsigned amask, unsigned emask) {
  using C = TileCfg<T, S>;
  constexpr int HB = (B - 1) / 2;
  T acc = T(0);
#pragma unroll
  for (int a = 0; a < B; ++a) {
#pragma unroll
    for (int e = 0; e < B; ++e) {
      const int ph = posmod(e - HB + OXM, S);
      const int dc = floordiv(e - HB + OXM, S);
      T v = xs[addr + a * C::XROW + ph * C::XPLANE + dc];
      if (BORDER) {
        // filter2D's BORDER_CONSTANT acts on the WARPED image: taps whose
        // (rr, cc) fall outside the H x W domain contribute 0
        v = (((amask >> a) & (emask >> e)) & 1u) ? v : T(0);
      }
      acc += blur[a * B + e] * v;
    }
  }
  return acc;
}

// Sweeps of the LR region by one wave: sweep j < lrh covers LR row j, lane =
// LR column (0..63); the tail sweep covers the remaining lrw - 64 (<= 3) columns
// of every row, lane -> (row, column).  No per-pixel division, row predicates
// are wave-uniform.
struct LaneTail {
  int li, lj;   // LR row / column of this lane in the tail sweep
  bool act;
};
__device__ __forceinline__ LaneTail tail_lane(int lane, int lrh, int lrw) {
  const int rem = lrw - 64;  // 1..3 (0 when the region is exactly 64 wide)
  LaneTail t;
  int li = lane, c = 0;
  if (rem == 2) { li = lane >> 1; c = lane & 1; }
  if (rem == 3) { li = (lane * 171) >> 9; c = lane - 3 * li; }
  t.li = li; t.lj = 64 + c;
  t.act = rem > 0 && li < lrh;
  if (!t.act) { t.li = 0; t.lj = 0; }
  return t;
}

// Observation of LR pixel (gi, gj) of one frame; addresses outside the LR image
// are clamped (the residual is masked to 0 later).
template <typename T, typename ArgsT>
__device__ __forceinline__ T load_obs(const ArgsT& A, const T* __restrict__ yk, int gi, int gj) {
  const bool ok = (unsigned)gi < (unsigned)A.hl && (unsigned)gj < (unsigned)A.wl;
  return yk[ok ? (size_t)gi * A.wl + gj : (size_t)0];
}

// Observations a wave needs first for its frame: LR rows 0 and 1 of the region
// and the tail sweep.  Issued a phase ahead of their use.
template <typename T>
struct ObsPrefetch { T y0, y1, yt; };

template <typename T, typename ArgsT>
__device__ __forceinline__ ObsPrefetch<T> prefetch_obs(const ArgsT& A, const T* __restrict__ yk, int lane,
                                                       const LaneTail& tl, int gi0, int gj0) {
  ObsPrefetch<T> o;
  o.y0 = load_obs<T>(A, yk, gi0, gj0 + lane);
  o.y1 = load_obs<T>(A, yk, gi0 + 1, gj0 + lane);
  o.yt = load_obs<T>(A, yk, gi0 + tl.li, gj0 + tl.lj);
  return o;
}

// One residual: stencil, minus observation, masks, cost, store.
template <typename T, int S, int B, int OXM, bool EDGE, typename ArgsT>
__device__ __forceinline__ void residual_one(const ArgsT& A, const T* __restrict__ xs, T* __restrict__ rsk, T yval,
                                             int li, int lj, int soff, int sy, int sx, bool act, bool valid,
                                             bool owned, unsigned amask, unsigned emask, double& cost_data) {
  using C = TileCfg<T, S>;
  const int addr = li * (S * C::XROW) + lj + soff;
  T res = forward_taps<T, S, B, OXM, EDGE>(xs, addr, A.blur, amask, emask) - yval;
  res = valid ? res : T(0);
  const double rd = owned ? (double)res : 0.0;  // each LR pixel is owned by exactly one tile
  cost_data += rd * rd;
  // stored frame-aligned (see FrameInfo::sy/sx); pixels the gather never reads are only costed
  const int ls = li - sy, lt = lj - sx;
  if (act && (unsigned)ls < (unsigned)C::GRH && (unsigned)lt < (unsigned)C::GRW) rsk[ls * C::GRW + lt] = res;
}

// ---- Phase B: residuals of ONE frame (this wave) over the tile's LR region ----
// One sweep per LR row (lane = LR column 0..63, row predicates wave-uniform)
// plus a tail sweep for columns 64..lrw-1.  The row loop is NOT unrolled (code
// size: the kernel must stay inside the instruction cache); the observation of
// row j+2 is loaded while row j is evaluated.  EDGE: the region contains LR row
// 0 or LR column 0, the only pixels whose blur taps reach outside the warped
// image (filter2D BORDER_CONSTANT), so only those tiles carry tap masks.
template <typename T, int S, int B, int OXM, bool EDGE, typename ArgsT>
__device__ __forceinline__ void residual_pass(const ArgsT& A, const T* __restrict__ xs, T* __restrict__ rsk,
                                              const T* __restrict__ yk, const ObsPrefetch<T>& op, int lane,
                                              const LaneTail& tl, int gi0, int gj0, int CI0, int CJ0, int soff,
                                              int sy, int sx, double& cost_data) {
  using C = TileCfg<T, S>;
  constexpr int HB = (B - 1) / 2;
  const int gj = gj0 + lane;
  const bool col_valid = (unsigned)gj < (unsigned)A.wl;
  const bool col_owned = (unsigned)(gj - CJ0) < (unsigned)C::CW;
  unsigned emask = 0xffffffffu;
  if (EDGE) {
    emask = 0;
#pragma unroll
    for (int e = 0; e < B; ++e) emask |= ((unsigned)(S * gj + e - HB) < (unsigned)A.W ? 1u : 0u) << e;
  }
  T y0 = op.y0, y1 = op.y1;
#pragma unroll 1
  for (int j = 0; j < A.lrh; ++j) {
    const T ycur = y0;
    y0 = y1;
    if (j + 2 < A.lrh) y1 = load_obs<T>(A, yk, gi0 + j + 2, gj);  // uniform branch
    const int gi = gi0 + j;
    const bool row_valid = (unsigned)gi < (unsigned)A.hl;      // uniform
    const bool row_owned = (unsigned)(gi - CI0) < (unsigned)C::CH && S * gi >= A.cr0 && S * gi < A.cr1;  // uniform
    unsigned amask = 0xffffffffu;
    if (EDGE) {
      amask = 0;
#pragma unroll
      for (int a = 0; a < B; ++a) amask |= ((unsigned)(S * gi + a - HB) < (unsigned)A.H ? 1u : 0u) << a;
    }
    residual_one<T, S, B, OXM, EDGE>(A, xs, rsk, ycur, j, lane, soff, sy, sx, true, row_valid && col_valid,
                                     row_owned && col_owned, amask, emask, cost_data);
  }
  if (A.lrw > 64) {  // uniform: tail columns 64..lrw-1 of every row
    const int gi = gi0 + tl.li, gjt = gj0 + tl.lj;
    const bool valid = tl.act && (unsigned)gi < (unsigned)A.hl && (unsigned)gjt < (unsigned)A.wl;
    const bool owned = tl.act && (unsigned)(gi - CI0) < (unsigned)C::CH && (unsigned)(gjt - CJ0) < (unsigned)C::CW &&
                       S * gi >= A.cr0 && S * gi < A.cr1;
    unsigned amask = 0xffffffffu, em = 0xffffffffu;
    if (EDGE) {
      amask = 0; em = 0;
#pragma unroll
      for (int a = 0; a < B; ++a) {
        amask |= ((unsigned)(S * gi + a - HB) < (unsigned)A.H ? 1u : 0u) << a;
        em |= ((unsigned)(S * gjt + a - HB) < (unsigned)A.W ? 1u : 0u) << a;
      }
    }
    residual_one<T, S, B, OXM, EDGE>(A, xs, rsk, op.yt, tl.li, tl.lj, soff, sy, sx, tl.act, valid, owned, amask, em,
                                     cost_data);
  }
}

template <typename T, int S, int B, bool EDGE, typename ArgsT>
__device__ __forceinline__ void residual_switch(const ArgsT& A, const T* xs, T* rsk, const T* yk,
                                                const ObsPrefetch<T>& op, int lane, const LaneTail& tl, int gi0,
                                                int gj0, int CI0, int CJ0, int soff, int sy, int sx, int fxm,
                                                double& cost_data) {
  if (fxm == 0) residual_pass<T, S, B, 0, EDGE>(A, xs, rsk, yk, op, lane, tl, gi0, gj0, CI0, CJ0, soff, sy, sx, cost_data);
  if (S >= 2 && fxm == 1) residual_pass<T, S, B, (S >= 2 ? 1 : 0), EDGE>(A, xs, rsk, yk, op, lane, tl, gi0, gj0, CI0, CJ0, soff, sy, sx, cost_data);
  if (S >= 3 && fxm == 2) residual_pass<T, S, B, (S >= 3 ? 2 : 0), EDGE>(A, xs, rsk, yk, op, lane, tl, gi0, gj0, CI0, CJ0, soff, sy, sx, cost_data);
  if (S >= 4 && fxm == 3) residual_pass<T, S, B, (S >= 4 ? 3 : 0), EDGE>(A, xs, rsk, yk, op, lane, tl, gi0, gj0, CI0, CJ0, soff, sy, sx, cost_data);
}

// ---- Phase C: gather of one frame into the S accumulators of one row thread ----
// Zero-insertion + blur^T + shift^T in gather form (image_model.cpp:93-101):
// with B <= S + 1 a pixel row receives from at most two LR rows and a pixel
// column from at most two LR columns, so the frame's contribution is
//   acc[pc] += sum_{dy,dx in {0,1}} wr[dy] * wc[pc][dx] * r_k[li + dy][lj + dx]
// with wave-uniform weights (1-D blur taps, or 0) that the host tabulates per
// (frame, row phase) and (frame, column phase): no branches, no per-phase code.
template <typename T, int S, bool BORDER>
__device__ __forceinline__ void gather_frame(T (&acc)[S], const T* __restrict__ rsb, T wr0, T wr1,
                                             const T* __restrict__ wc, unsigned cmask) {
  using C = TileCfg<T, S>;
  const T v00 = rsb[0], v01 = rsb[1], v10 = rsb[C::GRW], v11 = rsb[C::GRW + 1];
  const T t0 = wr0 * v00 + wr1 * v10;
  const T t1 = wr0 * v01 + wr1 * v11;
#pragma unroll
  for (int pc = 0; pc < S; ++pc) {
    T c = wc[2 * pc] * t0 + wc[2 * pc + 1] * t1;
    if (BORDER) c = ((cmask >> pc) & 1u) ? c : T(0);  // p' column outside the image -> 0
    acc[pc] += c;
  }
}

// ---- regulariser pass 1 for the S pixels of one row thread ----
// Stores 2*lambda*w*r (0 for pixels outside the image and, BTV only, for the
// absolute pixel (0,0), btv_regularizer.cpp:143-146) into cr.
template <typename T, int S, int REGK, int R, int NP, bool BORDER>
__device__ __forceinline__ void reg_pass1(T (&acc)[S], double& cost, const T* __restrict__ xs,
                                          T* __restrict__ cr, const T (&wv)[S], int xrow, int xcell, int crrow,
                                          int crcell, int gr, int gc0, int W, int H, T lambda,
                                          const T (&pw)[NP], bool cost_row = true) {
  using C = TileCfg<T, S>;
  constexpr int WIN = (REGK == 2) ? R : 1;  // taps extend WIN pixels right/down
  constexpr int NC = S + WIN;
  // The window is walked ROW BY ROW (one row of NC values live at a time, all S
  // pixels accumulate): same summation order per pixel as a pixel-by-pixel walk
  // (i outer, j inner), a third of the registers.
  T x0v[S], rv[S], dv[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) { rv[pc] = T(0); dv[pc] = T(0); }
#pragma unroll
  for (int i = 0; i <= WIN; ++i) {
    T row[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) row[j] = xs[(xrow + i) * C::XROW + (j % S) * C::XPLANE + xcell + j / S];
    if (i == 0) {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) x0v[pc] = row[pc];
    }
#pragma unroll
    for (int pc = 0; pc < S; ++pc) {
      if (REGK == 2) {
#pragma unroll
        for (int j = 0; j <= R; ++j) {
          if (i == 0 && j == 0) continue;  // |x0 - x0| = 0 and sgn(0) = 0
          T d = x0v[pc] - row[pc + j];
          if (BORDER) d = ((gr + i < H) && (gc0 + pc + j < W)) ? d : T(0);  // skipped tap == zero difference
          rv[pc] += pw[i + j] * absv(d);
          if (i < R && j < R) dv[pc] += sgn_scaled<T>(d, pw[i + j]);  // exclusive window in the gradient
        }
      } else if (i == 1) {
        // rv = |dy| + |dx|, dv = -sgn(dx) - sgn(dy) (tv_regularizer.cpp:154-170); row 0 handled below
        T dyv = row[pc] - x0v[pc];
        if (BORDER) dyv = (gr + 1 < H) ? dyv : T(0);
        rv[pc] = absv(dyv) + rv[pc];
        dv[pc] = dv[pc] - sgnv(dyv);
      } else {
        T dxv = row[pc + 1] - x0v[pc];
        if (BORDER) dxv = (gc0 + pc + 1 < W) ? dxv : T(0);
        rv[pc] = absv(dxv);
        dv[pc] = -sgnv(dxv);
      }
    }
  }
#pragma unroll
  for (int pc = 0; pc < S; ++pc) {
    const T r = rv[pc], didi = dv[pc];
    const T c = lambda * wv[pc];
    T cr2 = T(2) * c * r;
    acc[pc] += cr2 * didi;
    const bool in_img = gr < H && gc0 + pc < W;
    const double cd = (in_img && cost_row) ? (double)c * (double)r * (double)r : 0.0;
    cost += cd;
    if (!in_img || (REGK == 2 && gr == 0 && gc0 + pc == 0)) cr2 = T(0);
    cr[crrow * C::CRROW + pc * C::CRPLANE + crcell] = cr2;
  }
}

// ---- regulariser pass 1 for the up/left halo strips ----
// pass 2 reads 2*lambda*w*r of the RU pixel rows above and RU pixel columns left
// of the tile; they are recomputed here, one pixel per thread.  Task h < NTOP
// is a pixel of the top strip (rows -RU..-1, columns 0..TW-1: exactly RU * TW
// = a multiple of 64 tasks), the remaining RU * (TH + RU) tasks are the left
// strip including the corner.  The pixel's IRLS weight is prefetched at kernel
// start (halo_pixel + the caller), so no global latency sits in this phase.
template <typename T, int S, int REGK, int R>
struct HaloGeom {
  static constexpr int RU = (REGK == 2) ? R - 1 : (REGK == 1 ? 1 : 0);
  static constexpr int NTOP = RU * TileCfg<T, S>::TW;
  static constexpr int NH = NTOP + RU * (TileCfg<T, S>::TH + RU);
  static constexpr int NIT = NH > 0 ? (NH + TileCfg<T, S>::NT - 1) / TileCfg<T, S>::NT : 1;
};

template <typename T, int S, int REGK, int R>
__device__ __forceinline__ void halo_pixel(int h, int& row, int& col) {
  using G = HaloGeom<T, S, REGK, R>;
  using C = TileCfg<T, S>;
  if (h < G::NTOP) { row = h / C::TW - G::RU; col = h % C::TW; }
  else if (G::RU > 0) { const int h2 = h - G::NTOP; row = h2 / (G::RU > 0 ? G::RU : 1) - G::RU; col = h2 % (G::RU > 0 ? G::RU : 1) - G::RU; }
  else { row = 0; col = 0; }
}

template <typename T, int S, int REGK, int R, int NP>
__device__ __forceinline__ void reg_halo(const T* __restrict__ xs, T* __restrict__ cr,
                                         const T (&whalo)[HaloGeom<T, S, REGK, R>::NIT], int tid, int hu, int hlc,
                                         int R0, int C0, int W, int H, T lambda, const T (&pw)[NP]) {
  using C = TileCfg<T, S>;
  using G = HaloGeom<T, S, REGK, R>;
  if (G::RU == 0) return;
#pragma unroll
  for (int it = 0; it < G::NIT; ++it) {
    const int h = tid + it * C::NT;
    if (h >= G::NH) break;
    int row, col;  // tile-relative pixel coordinates (negative in the halo)
    halo_pixel<T, S, REGK, R>(h, row, col);
    const int gr = R0 + row, gc = C0 + col;
    T cr2 = T(0);
    if (gr >= 0 && gr < H && gc >= 0 && gc < W && !(REGK == 2 && gr == 0 && gc == 0)) {
      const int xr = hu + row, xc = col + hlc * S;  // >= 0 by construction of the plan
      const T x0 = xs[xr * C::XROW + (xc % S) * C::XPLANE + xc / S];
      T r = T(0);
      if (REGK == 2) {
#pragma unroll
        for (int j = 0; j <= R; ++j) {
          const int xcj = xc + j;
          const int cofs = (xcj % S) * C::XPLANE + xcj / S;
          const bool cin = gc + j < W;
#pragma unroll
          for (int i = 0; i <= R; ++i) {
            if (i == 0 && j == 0) continue;
            const T v = xs[(xr + i) * C::XROW + cofs];
            const T d = (cin && gr + i < H) ? x0 - v : T(0);
            r += pw[i + j] * absv(d);
          }
        }
      } else {
        const int xc1 = xc + 1;
        const T xv = (gc + 1 < W) ? absv(xs[xr * C::XROW + (xc1 % S) * C::XPLANE + xc1 / S] - x0) : T(0);
        const T yv = (gr + 1 < H) ? absv(xs[(xr + 1) * C::XROW + (xc % S) * C::XPLANE + xc / S] - x0) : T(0);
        r = yv + xv;
      }
      cr2 = T(2) * (lambda * whalo[it]) * r;
    }
    const int crr = row + G::RU, crc = col + S;
    cr[crr * C::CRROW + (crc % S) * C::CRPLANE + crc / S] = cr2;
  }
}

// ---- regulariser pass 2: contributions of the up/left neighbours ----
template <typename T, int S, int REGK, int R, int NP>
__device__ __forceinline__ void reg_pass2(T (&acc)[S], const T* __restrict__ xs, const T* __restrict__ cr,
                                          int xrow, int xcell, int crrow, int crcell, const T (&pw)[NP]) {
  using C = TileCfg<T, S>;
  constexpr int RU = (REGK == 2) ? R - 1 : 1;  // neighbours reach RU pixels up/left
  if (RU == 0) return;
  constexpr int NC = S + RU;
  // Row by row, farthest neighbour row last read = own row: the per-pixel sums
  // run i = 0 (own row) .. RU (farthest) as in a pixel-by-pixel walk, so the own
  // row is read first and kept (x0), then rows r-1, r-2, ...
  T x0v[S], sum[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) sum[pc] = T(0);
#pragma unroll
  for (int i = 0; i <= RU; ++i) {  // neighbour row r - i
    T xw[NC], cw[NC];              // columns -RU..S-1
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int col = j - RU, ph = posmod(col, S), dc = floordiv(col, S);
      xw[j] = xs[(xrow - i) * C::XROW + ph * C::XPLANE + xcell + dc];
      cw[j] = cr[(crrow - i) * C::CRROW + ph * C::CRPLANE + crcell + dc];
    }
    if (i == 0) {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) x0v[pc] = xw[pc + RU];
    }
#pragma unroll
    for (int pc = 0; pc < S; ++pc) {
      if (REGK == 2) {
        if (i < R) {
#pragma unroll
          for (int j = 0; j < R; ++j) {
            if (i == 0 && j == 0) continue;
            // -sgn(x[q] - x[p]) * alpha^(i+j) * 2 c[q] r[q],  q = p - (i, j)
            sum[pc] += cw[pc + RU - j] * sgn_scaled<T>(x0v[pc] - xw[pc + RU - j], pw[i + j]);
          }
        }
      } else {
        // left (i = 0) and above (i = 1) (tv_regularizer.cpp:172-203)
        if (i == 0) sum[pc] += cw[pc + RU - 1] * sgnv(x0v[pc] - xw[pc + RU - 1]);
        else sum[pc] += cw[pc + RU] * sgnv(x0v[pc] - xw[pc + RU]);
      }
    }
  }
#pragma unroll
  for (int pc = 0; pc < S; ++pc) acc[pc] += sum[pc];
}

}  // namespace
}  // namespace srmap
