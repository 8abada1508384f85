// kernels_tiled.hip -- the hot path: ONE fused kernel per MAP gradient
// iteration (ObjectiveFunction::ComputeAllTerms, objective_function.cpp:5-20)
// for the common geometry: integer motion shifts, HR = LR * S, S in {2,3,4},
// blur size B in {1,3}, and at most one regulariser handled in-kernel (2-D TV or
// BTV with range <= 3); anything else is evaluated by kernels_direct.hip.
//
// Design (DESIGN.md "Fused evaluation kernel"):
//   * A workgroup (256 threads, 4 waves) owns a tile of CH x CW LR cells = one
//     S x S block of HR pixels per thread ("cell-major"): every thread reads x /
//     IRLS weights and writes g as S-element vectors -> fully coalesced HBM
//     traffic, each compulsory byte crosses HBM once (x halo re-reads hit L2).
//   * The x tile (+halo) is staged in LDS in POLYPHASE layout
//     xs[row][column phase (c mod S)][cell]: the decimated forward stencil
//     (stride-S access) and the per-pixel regulariser windows both become
//     unit-stride across lanes -> no LDS bank conflicts, all tap offsets are
//     instruction immediates.
//   * Per chunk of 4 frames: wave w computes the LR residuals r_k = A_k x - y_k
//     of frame k0+w for the LR pixels the tile needs (warp -> blur -> decimate
//     fused, objective_data_term.cpp:27-50) into LDS; then every thread gathers
//     sum_k M_k^T B^T D^T r_k for its cell (image_model.cpp:93-101).  The
//     frame's shift phase (shift mod S) selects one of S*S fully unrolled code
//     paths by a wave-uniform switch, so tap positions and register targets
//     are compile-time and each valid tap costs one FMA.
//   * The regulariser (tv_regularizer.cpp:135-227 / btv_regularizer.cpp:93-170,
//     bug-compatible) runs on the same x tile: pass 1 computes r and c*r for the
//     tile plus an up/left halo into LDS and the self term, pass 2 adds the
//     neighbour terms.
//   * Cost partials (s^2 * sum r_k^2 and lambda * w * r^2) are reduced per
//     workgroup in fp64 with wave shuffles and written to a partials buffer that
//     the final one-block reduction sums in a fixed order (deterministic).
// No MFMA: this is a stencil/gather path.
#include <algorithm>
#include <climits>
#include <cstdlib>

#include "srmap_internal.hpp"

namespace srmap {

namespace {

constexpr int kThreads = 256;
constexpr int kFrameChunk = 4;   // frames per residual/gather round = waves per workgroup
constexpr int kMaxHaloRows = 12; // max hu + hd of the x tile
constexpr int kMaxHaloCells = 4; // max hlc + hrc

template <typename T, int S>
struct TileCfg {
  static constexpr int CW = 32;                       // LR cells per tile row
  static constexpr int CH = kThreads / CW;            // 8 cell rows
  static constexpr int TH = CH * S, TW = CW * S;      // HR tile
  static constexpr int XR = TH + kMaxHaloRows;        // x rows held in LDS
  static constexpr int XCELLS = CW + kMaxHaloCells;   // cells per x row
  static constexpr int XPLANE = XCELLS;               // elements per (row, phase)
  static constexpr int XROW = S * XPLANE;             // elements per row
  static constexpr int LRH = CH + 3, LRW = CW + 3;    // LR residual region (max)
  static constexpr int CRCELLS = CW + 1;              // c*r: one halo cell column
  static constexpr int CRPLANE = CRCELLS;
  static constexpr int CRROW = S * CRPLANE;
  static constexpr int CRR = TH + S;                  // one halo cell row
  static constexpr int XS_ELEMS = XR * XROW;
  static constexpr int RS_ELEMS = kFrameChunk * LRH * LRW;
  static constexpr int CR_ELEMS = CRR * CRROW;
  // rs (residuals) is dead once the gather is done; c*r reuses its space.
  static constexpr int SCRATCH_ELEMS = RS_ELEMS > CR_ELEMS ? RS_ELEMS : CR_ELEMS;
};

// Per-frame shift decomposition, precomputed on the host.
struct FrameInfo {
  int frow;   // forward: tile-row offset  S*i0 + oy + hu - hb   (x rows)
  int fcell;  // forward: cell offset      j0 + hlc + floor(ox / S)
  int fxm;    // forward: ox mod S  (0..S-1)
  int gbase;  // gather: (toyq - i0) * LRW + (toxq - j0)
  int gym;    // gather: toy mod S
  int gxm;    // gather: tox mod S
  int toy, tox;  // transpose integer offsets (for the border test)
};

template <typename T, int B, int NP>
struct FusedArgs {
  const T* x;
  const T* y;
  const T* w;       // IRLS weights or nullptr
  T* g;             // nullptr = cost only
  double* partials;
  unsigned* counter;   // arrival counter for the in-kernel final reduction (nullptr = off)
  double* cost_out;    // receives the total when counter != nullptr
  const FrameInfo* frames;
  int W, H, wl, hl, K;
  int obs_C, obs_c0;
  int hu, hlc;          // x tile origin = (R0 - hu, cell CJ0 - hlc)
  int xrows, xcells;    // loaded extent
  int i0, j0;           // LR region origin relative to the tile's first cell (<= 0)
  int lrh, lrw;         // LR region extent
  int margin;           // tiles closer than this to the image edge take the border path
  int terms;            // SRMAP_TERM_*
  T blur[B * B];        // k * k^T (blur_module.cpp:20-22)
  T lambda;
  T powtab[NP];         // BTV alpha^(i+j)
};

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

constexpr int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
constexpr int posmod(int a, int b) { return a - floordiv(a, b) * b; }

// sgn(d) * pw with sgn(0) = 0 (pw > 0).  f32: ldexp pushes every non-zero d
// (subnormals included) beyond pw, med3 clamps to +-pw: 2 VALU ops, exact.
template <typename T>
__device__ __forceinline__ T sgn_scaled(T d, T pw) { return d > T(0) ? pw : (d < T(0) ? -pw : T(0)); }
template <>
__device__ __forceinline__ float sgn_scaled<float>(float d, float pw) {
  return __builtin_amdgcn_fmed3f(__builtin_ldexpf(d, 200), -pw, pw);
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
// addr = element offset of xs[(S*li + 0 - hb + ...)][.][lj + ...] already
// including the frame's scalar offsets; taps use immediates.
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

// ---- gather of one frame into the S x S accumulators of one cell ----
// rsb points at rs[(lci + toyq - i0)][(lcj + toxq - j0)] of the frame.
template <typename T, int S, int B, int GYM, int GXM, bool BORDER>
__device__ __forceinline__ void gather_case(T (&acc)[S][S], const T* __restrict__ rsb,
                                            const T (&blur)[B * B], unsigned rmask, unsigned cmask) {
  using C = TileCfg<T, S>;
  constexpr int HB = (B - 1) / 2;
#pragma unroll
  for (int pr = 0; pr < S; ++pr) {
#pragma unroll
    for (int a = 0; a < B; ++a) {
      if (posmod(pr + GYM + a - HB, S) != 0) continue;  // zero-insertion: only multiples of S carry data
      const int dy = floordiv(pr + GYM + a - HB, S);
#pragma unroll
      for (int pc = 0; pc < S; ++pc) {
#pragma unroll
        for (int e = 0; e < B; ++e) {
          if (posmod(pc + GXM + e - HB, S) != 0) continue;
          const int dx = floordiv(pc + GXM + e - HB, S);
          T val = rsb[dy * C::LRW + dx];  // identical addresses are CSE'd by the compiler
          if (BORDER) {
            // warpAffine(-dx,-dy) samples v_k at p' = p + (toy, tox); outside -> 0
            val = (((rmask >> pr) & (cmask >> pc)) & 1u) ? val : T(0);
          }
          // kernel.t() (blur_module.cpp:35): Gt[a][e] = G[e][a]
          acc[pr][pc] += blur[e * B + a] * val;
        }
      }
    }
  }
}

template <typename T, int S, int B, int GYM, bool BORDER>
__device__ __forceinline__ void gather_switch_x(T (&acc)[S][S], const T* rsb, const T (&blur)[B * B],
                                                int gxm, unsigned rmask, unsigned cmask) {
  if (S >= 1 && gxm == 0) gather_case<T, S, B, GYM, 0, BORDER>(acc, rsb, blur, rmask, cmask);
  if (S >= 2 && gxm == 1) gather_case<T, S, B, GYM, (S >= 2 ? 1 : 0), BORDER>(acc, rsb, blur, rmask, cmask);
  if (S >= 3 && gxm == 2) gather_case<T, S, B, GYM, (S >= 3 ? 2 : 0), BORDER>(acc, rsb, blur, rmask, cmask);
  if (S >= 4 && gxm == 3) gather_case<T, S, B, GYM, (S >= 4 ? 3 : 0), BORDER>(acc, rsb, blur, rmask, cmask);
}

template <typename T, int S, int B, bool BORDER>
__device__ __forceinline__ void gather_switch(T (&acc)[S][S], const T* rsb, const T (&blur)[B * B],
                                              int gym, int gxm, unsigned rmask, unsigned cmask) {
  if (S >= 1 && gym == 0) gather_switch_x<T, S, B, 0, BORDER>(acc, rsb, blur, gxm, rmask, cmask);
  if (S >= 2 && gym == 1) gather_switch_x<T, S, B, (S >= 2 ? 1 : 0), BORDER>(acc, rsb, blur, gxm, rmask, cmask);
  if (S >= 3 && gym == 2) gather_switch_x<T, S, B, (S >= 3 ? 2 : 0), BORDER>(acc, rsb, blur, gxm, rmask, cmask);
  if (S >= 4 && gym == 3) gather_switch_x<T, S, B, (S >= 4 ? 3 : 0), BORDER>(acc, rsb, blur, gxm, rmask, cmask);
}

template <typename T, int S, int B, int OXMAX, bool BORDER>
__device__ __forceinline__ T forward_switch(const T* xs, int addr, const T (&blur)[B * B], int fxm,
                                            unsigned amask, unsigned emask) {
  T r = T(0);
  if (fxm == 0) r = forward_taps<T, S, B, 0, BORDER>(xs, addr, blur, amask, emask);
  if (S >= 2 && fxm == 1) r = forward_taps<T, S, B, (S >= 2 ? 1 : 0), BORDER>(xs, addr, blur, amask, emask);
  if (S >= 3 && fxm == 2) r = forward_taps<T, S, B, (S >= 3 ? 2 : 0), BORDER>(xs, addr, blur, amask, emask);
  if (S >= 4 && fxm == 3) r = forward_taps<T, S, B, (S >= 4 ? 3 : 0), BORDER>(xs, addr, blur, amask, emask);
  return r;
}

// Number of 64-lane sweeps that cover the largest LR residual region.
template <typename T, int S>
struct Sweep { static constexpr int MAXIT = (TileCfg<T, S>::LRH * TileCfg<T, S>::LRW + 63) / 64; };

// Per-lane description of the LR pixels one lane visits in its 64-lane sweeps
// over the tile's LR region.  Frame-invariant, so it is decoded once per
// kernel: bits 0-7 lj, 8-15 li, 16 valid (inside the LR image), 17 owned by
// this tile, 18 active (inside the region), 19.. warped-domain tap masks
// (B row bits, then B column bits; only used by border tiles).
template <typename T, int S, int B, typename ArgsT>
__device__ __forceinline__ void build_sweep(const ArgsT& A, int lane, int CI0, int CJ0,
                                            unsigned (&tab)[Sweep<T, S>::MAXIT]) {
  using C = TileCfg<T, S>;
  constexpr int HB = (B - 1) / 2;
  const int nlr = A.lrh * A.lrw;
  const float invw = 1.0f / (float)A.lrw;
  const int gi0 = CI0 + A.i0, gj0 = CJ0 + A.j0;
#pragma unroll
  for (int it = 0; it < Sweep<T, S>::MAXIT; ++it) {
    const int nidx = lane + 64 * it;
    const bool act = nidx < nlr;
    const int idx = act ? nidx : nlr - 1;
    const int li = (int)(((float)idx + 0.5f) * invw), lj = idx - li * A.lrw;
    const int gi = gi0 + li, gj = gj0 + lj;
    const bool valid = (unsigned)gi < (unsigned)A.hl && (unsigned)gj < (unsigned)A.wl;
    const bool owned = act && (unsigned)(gi - CI0) < (unsigned)C::CH && (unsigned)(gj - CJ0) < (unsigned)C::CW;
    unsigned v = (unsigned)lj | ((unsigned)li << 8) | (valid ? 1u << 16 : 0u) | (owned ? 1u << 17 : 0u) |
                 (act ? 1u << 18 : 0u);
#pragma unroll
    for (int a = 0; a < B; ++a) {
      const int rr = S * gi + a - HB, cc = S * gj + a - HB;
      v |= ((unsigned)rr < (unsigned)A.H ? 1u : 0u) << (19 + a);
      v |= ((unsigned)cc < (unsigned)A.W ? 1u : 0u) << (19 + B + a);
    }
    tab[it] = v;
  }
}

// Issue the observation loads of one frame for the whole LR region (no waits:
// the values are consumed a full gather phase later).
template <typename T, int S, typename ArgsT>
__device__ __forceinline__ void load_observations(const ArgsT& A, const T* __restrict__ yk, int CI0, int CJ0,
                                                  const unsigned (&tab)[Sweep<T, S>::MAXIT],
                                                  T (&yv)[Sweep<T, S>::MAXIT]) {
  const int gi0 = CI0 + A.i0, gj0 = CJ0 + A.j0;
#pragma unroll
  for (int it = 0; it < Sweep<T, S>::MAXIT; ++it) {
    const unsigned v = tab[it];
    const int li = (v >> 8) & 0xff, lj = v & 0xff;
    const bool valid = (v >> 16) & 1u;
    yv[it] = yk[valid ? (size_t)(gi0 + li) * A.wl + (gj0 + lj) : (size_t)0];  // clamped address, masked later
  }
}

// ---- Phase B body: residuals of ONE frame (wave-uniform) for the LR region ----
// Branch-free per lane (predicates become selects); observations were loaded
// a phase earlier.
template <typename T, int S, int B, int OXM, bool BORDER, typename ArgsT>
__device__ __forceinline__ void residual_pass(const ArgsT& A, const T* __restrict__ xs, T* __restrict__ rsk,
                                              const unsigned (&tab)[Sweep<T, S>::MAXIT],
                                              const T (&yv)[Sweep<T, S>::MAXIT], int frow, int fcell,
                                              double& cost_data) {
  using C = TileCfg<T, S>;
  const int nit = (A.lrh * A.lrw + 63) >> 6;
  const int soff = frow * C::XROW + fcell;  // frame-dependent part of the tap address (scalar)
#pragma unroll
  for (int it = 0; it < Sweep<T, S>::MAXIT; ++it) {
    if (it < nit) {  // uniform
      const unsigned v = tab[it];
      const int li = (v >> 8) & 0xff, lj = v & 0xff;
      const int addr = li * (S * C::XROW) + lj + soff;
      const unsigned amask = BORDER ? (v >> 19) : 0xffffffffu;
      const unsigned emask = BORDER ? (v >> (19 + B)) : 0xffffffffu;
      T res = forward_taps<T, S, B, OXM, BORDER>(xs, addr, A.blur, amask, emask) - yv[it];
      res = ((v >> 16) & 1u) ? res : T(0);
      // each LR pixel is owned by exactly one tile
      const double rd = ((v >> 17) & 1u) ? (double)res : 0.0;
      cost_data += rd * rd;
      if ((v >> 18) & 1u) rsk[li * C::LRW + lj] = res;
    }
  }
}

template <typename T, int S, int B, bool BORDER, typename ArgsT>
__device__ __forceinline__ void residual_switch(const ArgsT& A, const T* xs, T* rsk,
                                                const unsigned (&tab)[Sweep<T, S>::MAXIT],
                                                const T (&yv)[Sweep<T, S>::MAXIT], int frow, int fcell, int fxm,
                                                double& cost_data) {
  if (fxm == 0) residual_pass<T, S, B, 0, BORDER>(A, xs, rsk, tab, yv, frow, fcell, cost_data);
  if (S >= 2 && fxm == 1) residual_pass<T, S, B, (S >= 2 ? 1 : 0), BORDER>(A, xs, rsk, tab, yv, frow, fcell, cost_data);
  if (S >= 3 && fxm == 2) residual_pass<T, S, B, (S >= 3 ? 2 : 0), BORDER>(A, xs, rsk, tab, yv, frow, fcell, cost_data);
  if (S >= 4 && fxm == 3) residual_pass<T, S, B, (S >= 4 ? 3 : 0), BORDER>(A, xs, rsk, tab, yv, frow, fcell, cost_data);
}

// ---- regulariser pass 1 for one cell: values r, c*r products, self term ----
// cell at tile-relative cell coords (ci, cj) in [-1, CH) x [-1, CW); xcell/xrow
// locate its first pixel in xs.  Stores 2*c*r (0 for pixels outside the image
// and for the absolute pixel (0,0), btv_regularizer.cpp:143-146) into cr.
template <typename T, int S, int REGK, int R, int NP, bool BORDER, bool OWNED>
__device__ __forceinline__ void reg_pass1(T (&acc)[S][S], double& cost, const T* __restrict__ xs,
                                          T* __restrict__ cr, const T (&wv)[S][S],
                                          int xrow0, int xcell0, int crrow0, int crcell0, int gr0, int gc0,
                                          int W, int H, T lambda, const T (&pw)[NP]) {
  using C = TileCfg<T, S>;
  constexpr int WIN = (REGK == 2) ? R : 1;  // taps extend WIN pixels right/down
  constexpr int NC = S + WIN;               // columns of x needed per row
  T win[WIN + 1][NC];
  // preload rows 0..WIN-1 of the rolling window
#pragma unroll
  for (int i = 0; i < WIN; ++i)
#pragma unroll
    for (int j = 0; j < NC; ++j)
      win[i][j] = xs[(xrow0 + i) * C::XROW + (j % S) * C::XPLANE + xcell0 + j / S];
#pragma unroll
  for (int pr = 0; pr < S; ++pr) {
    // row (pr + WIN) enters the window at slot (pr + WIN) % (WIN + 1)
#pragma unroll
    for (int j = 0; j < NC; ++j)
      win[(pr + WIN) % (WIN + 1)][j] = xs[(xrow0 + pr + WIN) * C::XROW + (j % S) * C::XPLANE + xcell0 + j / S];
    const int gr = gr0 + pr;
#pragma unroll
    for (int pc = 0; pc < S; ++pc) {
      const T x0 = win[pr % (WIN + 1)][pc];
      T r = T(0), didi = T(0);
      if (REGK == 2) {
#pragma unroll
        for (int i = 0; i <= R; ++i) {
#pragma unroll
          for (int j = 0; j <= R; ++j) {
            if (i == 0 && j == 0) continue;  // |x0 - x0| = 0 and sgn(0) = 0
            const T d = x0 - win[(pr + i) % (WIN + 1)][pc + j];
            bool inside = true;
            if (BORDER) inside = (gr + i < H) && (gc0 + pc + j < W);
            if (inside) {
              r += pw[i + j] * absv(d);
              if (i < R && j < R) didi += sgn_scaled<T>(d, pw[i + j]);  // exclusive window in the gradient
            }
          }
        }
      } else {
        const T dyv = win[(pr + 1) % (WIN + 1)][pc] - x0;
        const T dxv = win[pr % (WIN + 1)][pc + 1] - x0;
        bool iny = true, inx = true;
        if (BORDER) { iny = gr + 1 < H; inx = gc0 + pc + 1 < W; }
        const T yv = iny ? absv(dyv) : T(0);
        const T xv = inx ? absv(dxv) : T(0);
        r = yv + xv;
        if (inx) didi -= sgnv(dxv);
        if (iny) didi -= sgnv(dyv);
      }
      const T c = lambda * wv[pr][pc];
      T cr2 = T(2) * c * r;
      const bool in_img = gr >= 0 && gr < H && gc0 + pc >= 0 && gc0 + pc < W;
      if (OWNED) {
        acc[pr][pc] += cr2 * didi;
        if (in_img) cost += (double)c * (double)r * (double)r;
      }
      // BTV only: the absolute pixel (0,0) never back-propagates (btv_regularizer.cpp:143-146)
      if (!in_img || (REGK == 2 && gr == 0 && gc0 + pc == 0)) cr2 = T(0);
      cr[(crrow0 + pr) * C::CRROW + pc * C::CRPLANE + crcell0] = cr2;
    }
  }
}

// ---- regulariser pass 2: contributions of up/left neighbours ----
template <typename T, int S, int REGK, int R, int NP>
__device__ __forceinline__ void reg_pass2(T (&acc)[S][S], const T* __restrict__ xs,
                                          const T* __restrict__ cr, int xrow0, int xcell0, int crrow0,
                                          int crcell0, const T (&pw)[NP]) {
  using C = TileCfg<T, S>;
  constexpr int RU = (REGK == 2) ? R - 1 : 1;  // neighbours reach RU pixels up/left
  if (RU == 0) return;
  constexpr int NC = S + RU;
  // windows cover rows (pr - RU .. pr), columns (-RU .. S-1) of the cell
  T xw[RU + 1][NC], cw[RU + 1][NC];
#pragma unroll
  for (int i = 0; i < RU; ++i) {  // rows -RU .. -1 relative to the cell
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int col = j - RU, ph = posmod(col, S), dc = floordiv(col, S);
      xw[posmod(i - RU, RU + 1)][j] = xs[(xrow0 + i - RU) * C::XROW + ph * C::XPLANE + xcell0 + dc];
      cw[posmod(i - RU, RU + 1)][j] = cr[(crrow0 + i - RU) * C::CRROW + ph * C::CRPLANE + crcell0 + dc];
    }
  }
#pragma unroll
  for (int pr = 0; pr < S; ++pr) {
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int col = j - RU, ph = posmod(col, S), dc = floordiv(col, S);
      xw[pr % (RU + 1)][j] = xs[(xrow0 + pr) * C::XROW + ph * C::XPLANE + xcell0 + dc];
      cw[pr % (RU + 1)][j] = cr[(crrow0 + pr) * C::CRROW + ph * C::CRPLANE + crcell0 + dc];
    }
#pragma unroll
    for (int pc = 0; pc < S; ++pc) {
      const T x0 = xw[pr % (RU + 1)][pc + RU];
      T sum = T(0);
      if (REGK == 2) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
#pragma unroll
          for (int j = 0; j < R; ++j) {
            if (i == 0 && j == 0) continue;
            const T xq = xw[posmod(pr - i, RU + 1)][pc + RU - j];
            const T cq = cw[posmod(pr - i, RU + 1)][pc + RU - j];
            sum += cq * sgn_scaled<T>(x0 - xq, pw[i + j]);  // -sgn(x[q]-x[p]) * alpha^(i+j)
          }
        }
      } else {
        // left and above (tv_regularizer.cpp:172-203)
        sum += cw[pr % (RU + 1)][pc + RU - 1] * sgnv(x0 - xw[pr % (RU + 1)][pc + RU - 1]);
        sum += cw[posmod(pr - 1, RU + 1)][pc + RU] * sgnv(x0 - xw[posmod(pr - 1, RU + 1)][pc + RU]);
      }
      acc[pr][pc] += sum;
    }
  }
}

// ---- regulariser pass 1 for the up/left halo strips ----
// pass 2 reads c*r of the RU pixel rows above and RU pixel columns left of the
// tile; they are recomputed here, one pixel per thread, spread over the whole
// workgroup (RU*(TW+RU) + RU*TH pixels).
template <typename T, int S, int REGK, int R, int NP>
__device__ __forceinline__ void reg_halo(const T* __restrict__ xs, T* __restrict__ cr,
                                         const T* __restrict__ wplane, int tid, int hu, int hlc, int R0,
                                         int C0, int W, int H, T lambda, const T (&pw)[NP]) {
  using C = TileCfg<T, S>;
  constexpr int RU = (REGK == 2) ? R - 1 : 1;
  if (RU == 0) return;
  constexpr int TOPW = C::TW + RU;
  constexpr int NTOP = RU * TOPW, NH = NTOP + RU * C::TH;
  for (int h = tid; h < NH; h += kThreads) {
    int row, col;  // tile-relative pixel coordinates (negative in the halo)
    if (h < NTOP) { row = h / TOPW - RU; col = h % TOPW - RU; }
    else { const int h2 = h - NTOP; row = h2 / RU; col = h2 % RU - RU; }
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
            if (cin && gr + i < H) r += pw[i + j] * absv(x0 - v);
          }
        }
      } else {
        const int xc1 = xc + 1;
        const T xv = (gc + 1 < W) ? absv(xs[xr * C::XROW + (xc1 % S) * C::XPLANE + xc1 / S] - x0) : T(0);
        const T yv = (gr + 1 < H) ? absv(xs[(xr + 1) * C::XROW + (xc % S) * C::XPLANE + xc / S] - x0) : T(0);
        r = yv + xv;
      }
      const T wv = wplane ? wplane[(size_t)gr * W + gc] : T(1);
      cr2 = T(2) * (lambda * wv) * r;
    }
    const int crr = row + S, crc = col + S;
    cr[crr * C::CRROW + (crc % S) * C::CRPLANE + crc / S] = cr2;
  }
}

template <typename T, int S, int B, int REGK, int R>
__global__ __launch_bounds__(kThreads) void k_eval_fused(
    FusedArgs<T, B, (REGK == 2 ? 2 * R + 1 : 1)> A) {
  using C = TileCfg<T, S>;
  constexpr int NP = (REGK == 2 ? 2 * R + 1 : 1);
  constexpr int HB = (B - 1) / 2;
  __shared__ T xs[C::XS_ELEMS];
  __shared__ T scratch[C::SCRATCH_ELEMS];
  __shared__ double red[2][4];
  T* rs = scratch;
  T* cr = scratch;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index as an SGPR
  const int CI0 = blockIdx.y * C::CH, CJ0 = blockIdx.x * C::CW;
  const int R0 = CI0 * S, C0 = CJ0 * S;
  const int ch = blockIdx.z;
  const size_t N = (size_t)A.W * A.H;
  const T* xplane = A.x + (size_t)ch * N;
  const bool border = (R0 < A.margin) || (R0 + C::TH + A.margin > A.H) || (C0 < A.margin) ||
                      (C0 + C::TW + A.margin > A.W);

  const int lci = tid / C::CW, lcj = tid - lci * C::CW;  // this thread's cell
  const size_t nl = (size_t)A.wl * A.hl;
  constexpr int MAXIT = Sweep<T, S>::MAXIT;

  // ---------------- prefetch: every global load whose address is known now ----------------
  // HBM/L2 latency (~1 us) is as long as a whole phase of this kernel, so loads
  // are issued as early as possible and consumed phases later.
  unsigned sweep[MAXIT];  // this lane's LR pixels in the residual sweeps
  T yv[MAXIT];            // observations of frame (round 0, this wave)
  if (A.terms & SRMAP_TERM_DATA) {
    build_sweep<T, S, B>(A, lane, CI0, CJ0, sweep);
    if (wv < A.K) load_observations<T, S>(A, A.y + ((size_t)wv * A.obs_C + ch + A.obs_c0) * nl, CI0, CJ0, sweep, yv);
  }
  T wreg[S][S];  // IRLS weights of this thread's cell
  if (REGK != 0 && (A.terms & SRMAP_TERM_REG)) {
    const T* wplane = A.w ? A.w + (size_t)ch * N : nullptr;
    const int gc0 = C0 + S * lcj;
#pragma unroll
    for (int pr = 0; pr < S; ++pr) {
      const int gr = R0 + S * lci + pr;
      const bool in = wplane != nullptr && gr < A.H && gc0 < A.W;
#pragma unroll
      for (int pc = 0; pc < S; ++pc) wreg[pr][pc] = in ? wplane[(size_t)gr * A.W + gc0 + pc] : T(1);
    }
  }

  // ---------------- Phase A: x tile (+halo) -> LDS, polyphase ----------------
  {
    constexpr int AIT = (C::XR * C::XCELLS + kThreads - 1) / kThreads;
    const int total = (A.terms & 0x400) ? 0 : A.xrows * A.xcells;
    const float inv = 1.0f / (float)A.xcells;
    T vals[AIT][S];
    // stage 1: all loads in flight
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      const int idx = tid + it * kThreads;
      const int row = (int)(((float)idx + 0.5f) * inv);
      const int cell = idx - row * A.xcells;
      const int gr = R0 - A.hu + row;
      const int gcell = CJ0 - A.hlc + cell;
      const bool in = idx < total && (unsigned)gr < (unsigned)A.H && (unsigned)gcell < (unsigned)A.wl;
      const T* src = xplane + (in ? (size_t)gr * A.W + (size_t)gcell * S : (size_t)0);
#pragma unroll
      for (int pc = 0; pc < S; ++pc) vals[it][pc] = src[pc];
#pragma unroll
      for (int pc = 0; pc < S; ++pc) vals[it][pc] = in ? vals[it][pc] : T(0);
    }
    // stage 2: polyphase scatter into LDS
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      const int idx = tid + it * kThreads;
      if (idx < total) {
        const int row = (int)(((float)idx + 0.5f) * inv);
        const int cell = idx - row * A.xcells;
#pragma unroll
        for (int pc = 0; pc < S; ++pc) xs[row * C::XROW + pc * C::XPLANE + cell] = vals[it][pc];
      }
    }
  }
  __syncthreads();

  T acc[S][S];
#pragma unroll
  for (int i = 0; i < S; ++i)
#pragma unroll
    for (int j = 0; j < S; ++j) acc[i][j] = T(0);
  double cost_data = 0.0, cost_reg = 0.0;

  if (A.terms & SRMAP_TERM_DATA) {
    for (int k0 = 0; k0 < A.K; k0 += kFrameChunk) {
      // ---------------- Phase B: residuals of frame k0 + wave ----------------
      const int k = k0 + wv;  // wave-uniform (wv comes from readfirstlane)
      if (k < A.K && !(A.terms & 0x100)) {
        // scalar loads: the frame descriptor lives in SGPRs, the phase switch
        // below is a uniform branch
        const int frow = A.frames[k].frow, fcell = A.frames[k].fcell, fxm = A.frames[k].fxm;
        T* rsk = rs + wv * (C::LRH * C::LRW);
        if (border) residual_switch<T, S, B, true>(A, xs, rsk, sweep, yv, frow, fcell, fxm, cost_data);
        else residual_switch<T, S, B, false>(A, xs, rsk, sweep, yv, frow, fcell, fxm, cost_data);
      }
      // observations of the next round: in flight during the gather
      if (k + kFrameChunk < A.K)
        load_observations<T, S>(A, A.y + ((size_t)(k + kFrameChunk) * A.obs_C + ch + A.obs_c0) * nl, CI0, CJ0, sweep, yv);
      __syncthreads();
      // ---------------- Phase C: gather into this thread's cell ----------------
      if (A.g != nullptr && !(A.terms & 0x200)) {
        const int kc = (A.K - k0) < kFrameChunk ? (A.K - k0) : kFrameChunk;
        // all scalar loads of the round first: one exposed latency per round
        int gb[kFrameChunk], gym[kFrameChunk], gxm[kFrameChunk], toy[kFrameChunk], tox[kFrameChunk];
#pragma unroll
        for (int kk = 0; kk < kFrameChunk; ++kk) {
          const int kq = (k0 + kk < A.K) ? k0 + kk : A.K - 1;
          gb[kk] = A.frames[kq].gbase; gym[kk] = A.frames[kq].gym; gxm[kk] = A.frames[kq].gxm;
          toy[kk] = border ? A.frames[kq].toy : 0; tox[kk] = border ? A.frames[kq].tox : 0;
        }
#pragma unroll
        for (int kk = 0; kk < kFrameChunk; ++kk) {
          if (kk < kc) {
            const T* rsb = rs + kk * (C::LRH * C::LRW) + lci * C::LRW + lcj + gb[kk];
            if (border) {
              unsigned rmask = 0, cmask = 0;
#pragma unroll
              for (int q = 0; q < S; ++q) {
                const int pr_ = R0 + S * lci + q + toy[kk], pc_ = C0 + S * lcj + q + tox[kk];
                rmask |= ((unsigned)pr_ < (unsigned)A.H ? 1u : 0u) << q;
                cmask |= ((unsigned)pc_ < (unsigned)A.W ? 1u : 0u) << q;
              }
              gather_switch<T, S, B, true>(acc, rsb, A.blur, gym[kk], gxm[kk], rmask, cmask);
            } else {
              gather_switch<T, S, B, false>(acc, rsb, A.blur, gym[kk], gxm[kk], 0xffffffffu, 0xffffffffu);
            }
          }
        }
      }
      __syncthreads();
    }
    const T sc = (T)(2 * S * S);  // g += 2 * (s*s block sum) (objective_data_term.cpp:55-71)
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
      for (int j = 0; j < S; ++j) acc[i][j] *= sc;
  }

  // ---------------- Phase D: regulariser ----------------
  if (REGK != 0 && (A.terms & SRMAP_TERM_REG)) {
    const T* wplane = A.w ? A.w + (size_t)ch * N : nullptr;
    // pass 1: owned cell (weights were prefetched at kernel start)
    {
      const int xrow0 = A.hu + S * lci, xcell0 = A.hlc + lcj;
      if (border)
        reg_pass1<T, S, REGK, R, NP, true, true>(acc, cost_reg, xs, cr, wreg, xrow0, xcell0, S * (lci + 1),
                                                 lcj + 1, R0 + S * lci, C0 + S * lcj, A.W, A.H, A.lambda, A.powtab);
      else
        reg_pass1<T, S, REGK, R, NP, false, true>(acc, cost_reg, xs, cr, wreg, xrow0, xcell0, S * (lci + 1),
                                                  lcj + 1, R0 + S * lci, C0 + S * lcj, A.W, A.H, A.lambda, A.powtab);
    }
    // pass 1 for the halo strips (needed by pass 2 only)
    if (A.g != nullptr)
      reg_halo<T, S, REGK, R, NP>(xs, cr, wplane, tid, A.hu, A.hlc, R0, C0, A.W, A.H, A.lambda, A.powtab);
    __syncthreads();
    if (A.g != nullptr)
      reg_pass2<T, S, REGK, R, NP>(acc, xs, cr, A.hu + S * lci, A.hlc + lcj, S * (lci + 1), lcj + 1, A.powtab);
  }

  // ---------------- write g (S-element rows per thread, coalesced) ----------------
  if (A.g != nullptr) {
    const int gcj = CJ0 + lcj;
    if (gcj < A.wl) {
#pragma unroll
      for (int pr = 0; pr < S; ++pr) {
        const int gr = R0 + S * lci + pr;
        if (gr < A.H) {
          T* dst = A.g + (size_t)ch * N + (size_t)gr * A.W + (size_t)gcj * S;
#pragma unroll
          for (int pc = 0; pc < S; ++pc) dst[pc] = acc[pr][pc];
        }
      }
    }
  }

  // ---------------- cost partials ----------------
  {
    const double sd = wave_sum_d(cost_data);
    const double sr = wave_sum_d(cost_reg);
    if (lane == 0) { red[0][wv] = sd; red[1][wv] = sr; }
    __syncthreads();
    const unsigned nblocks = gridDim.x * gridDim.y * gridDim.z;
    if (tid == 0) {
      const double d = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
      const double r = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
      const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      const double part = (double)(S * S) * d + r;
      if (A.counter == nullptr) {
        A.partials[b] = part;
      } else {
        // publish write-through (sc1), drain, then take a ticket: the last
        // arriver sums all partials in index order -> deterministic total
        // without a second launch (cdna guide, G16 "R1" form)
        __hip_atomic_store(&A.partials[b], part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned ticket = __hip_atomic_fetch_add(A.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        red[0][0] = (ticket == nblocks - 1) ? 1.0 : 0.0;
      }
    }
    if (A.counter != nullptr) {
      __syncthreads();
      if (red[0][0] != 0.0) {  // workgroup-uniform: this is the last workgroup
        __syncthreads();
        double v = 0.0;
        for (unsigned i = tid; i < nblocks; i += kThreads)
          v += __hip_atomic_load(&A.partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = wave_sum_d(v);
        if (lane == 0) red[1][wv] = v;
        __syncthreads();
        if (tid == 0) {
          A.cost_out[0] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
          __hip_atomic_store(A.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        }
      }
    }
  }
}

// integer floor division / modulo on the host
inline int fdiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
inline int pmod(int a, int b) { return a - fdiv(a, b) * b; }

struct HostPlan {
  bool ok = false;
  int hu = 0, hd = 0, hlc = 0, hrc = 0, i0 = 0, j0 = 0, lrh = 0, lrw = 0, margin = 0;
  int regk = 0, regr = 0, reg_index = -1;
  std::vector<FrameInfo> frames;
};

// Decide whether the fused kernel covers the problem and derive the tile halos.
template <int S>
static HostPlan make_plan(const srmap_problem* p, int CH, int CW) {
  HostPlan pl;
  const Geometry& g = p->geo;
  const int B = g.b, hb = g.hb, K = g.K;
  if (!p->maps_regular) return pl;
  if (B != 1 && B != 3) return pl;
  // integer shifts only
  std::vector<int> ox(K, 0), oy(K, 0), tx(K, 0), ty(K, 0);
  if (p->has_motion) {
    for (int k = 0; k < K; ++k) {
      if (p->fwd_warps[k].ntaps != 1 || p->bwd_warps[k].ntaps != 1) return pl;
      ox[k] = p->fwd_warps[k].ox; oy[k] = p->fwd_warps[k].oy;
      tx[k] = p->bwd_warps[k].ox; ty[k] = p->bwd_warps[k].oy;
    }
  }
  // the one regulariser handled in-kernel (first TV / BTV with lambda > 0)
  for (int r = 0; r < p->nreg && pl.regk == 0; ++r) {
    const RegSpec& rs = p->reg[r];
    if (rs.lambda <= 0) continue;
    if (rs.kind == SRMAP_REG_TV) { pl.regk = 1; pl.reg_index = r; }
    else if (rs.kind == SRMAP_REG_BTV && rs.range >= 1 && rs.range <= 3 && rs.range - 1 <= S) {
      pl.regk = 2; pl.regr = rs.range; pl.reg_index = r;
    }
    break;  // only the first active regulariser may be fused (order of accumulation)
  }
  // LR region needed by the gather (+ the owned pixels)
  int dimin = 0, dimax = 0, djmin = 0, djmax = 0;
  int amax = 0;
  for (int k = 0; k < K; ++k) {
    dimin = std::min(dimin, -fdiv(-(ty[k] - hb), S));            // ceil((toy-hb)/S)
    dimax = std::max(dimax, fdiv(ty[k] + S - 1 + B - 1 - hb, S));
    djmin = std::min(djmin, -fdiv(-(tx[k] - hb), S));
    djmax = std::max(djmax, fdiv(tx[k] + S - 1 + B - 1 - hb, S));
    amax = std::max(amax, std::max(std::abs(ty[k]), std::abs(tx[k])));
    amax = std::max(amax, std::max(std::abs(oy[k]), std::abs(ox[k])));
  }
  pl.i0 = dimin; pl.j0 = djmin;
  pl.lrh = CH + dimax - dimin; pl.lrw = CW + djmax - djmin;
  if (pl.lrh > CH + 3 || pl.lrw > CW + 3) return pl;
  // x rows/cols the forward model touches for those LR pixels (tile-relative)
  int rmin = 0, rmax = CH * S - 1, cmin = 0, cmax = CW * S - 1;
  for (int k = 0; k < K; ++k) {
    rmin = std::min(rmin, S * dimin - hb + oy[k]);
    rmax = std::max(rmax, S * (CH - 1 + dimax) + (B - 1 - hb) + oy[k]);
    cmin = std::min(cmin, S * djmin - hb + ox[k]);
    cmax = std::max(cmax, S * (CW - 1 + djmax) + (B - 1 - hb) + ox[k]);
  }
  if (pl.regk) {
    const int win = pl.regk == 2 ? pl.regr : 1;
    rmin = std::min(rmin, -S);                      // halo cell row / column
    cmin = std::min(cmin, -S);
    rmax = std::max(rmax, CH * S - 1 + win);
    cmax = std::max(cmax, CW * S - 1 + win);
  }
  pl.hu = -rmin;
  pl.hd = rmax - (CH * S - 1);
  pl.hlc = -fdiv(cmin, S);
  pl.hrc = fdiv(cmax, S) - (CW - 1);
  if (pl.hu + pl.hd > kMaxHaloRows || pl.hlc + pl.hrc > kMaxHaloCells) return pl;
  pl.margin = std::max(std::max(pl.hu, pl.hd), std::max(pl.hlc, pl.hrc) * S) + amax + hb + 4;
  pl.frames.resize(K);
  for (int k = 0; k < K; ++k) {
    FrameInfo& f = pl.frames[k];
    f.frow = S * pl.i0 + oy[k] + pl.hu - hb;
    f.fcell = pl.j0 + pl.hlc + fdiv(ox[k], S);
    f.fxm = pmod(ox[k], S);
    f.gbase = (fdiv(ty[k], S) - pl.i0) * (CW + 3) + (fdiv(tx[k], S) - pl.j0);
    f.gym = pmod(ty[k], S);
    f.gxm = pmod(tx[k], S);
    f.toy = ty[k]; f.tox = tx[k];
  }
  pl.ok = true;
  return pl;
}

struct PlanCache {
  HostPlan plan;
  FrameInfo* d_frames = nullptr;
};

}  // namespace

// The plan is rebuilt whenever the regulariser list changes (cheap, host only)
// and cached on the problem through an opaque pointer table.
static std::vector<std::pair<const srmap_problem*, PlanCache>>& plan_table() {
  static std::vector<std::pair<const srmap_problem*, PlanCache>> t;
  return t;
}

static PlanCache* find_plan(const srmap_problem* p) {
  for (auto& e : plan_table()) if (e.first == p) return &e.second;
  return nullptr;
}

void tiled_release(srmap_problem* p) {
  auto& t = plan_table();
  for (size_t i = 0; i < t.size(); ++i)
    if (t[i].first == p) {
      if (t[i].second.d_frames) (void)hipFree(t[i].second.d_frames);
      t.erase(t.begin() + i);
      return;
    }
}

bool tiled_plan(srmap_problem* p) {
  tiled_release(p);
  const int S = p->geo.s;
  HostPlan pl;
  if (S == 2) pl = make_plan<2>(p, TileCfg<float, 2>::CH, TileCfg<float, 2>::CW);
  else if (S == 3) pl = make_plan<3>(p, TileCfg<float, 3>::CH, TileCfg<float, 3>::CW);
  else if (S == 4) pl = make_plan<4>(p, TileCfg<float, 4>::CH, TileCfg<float, 4>::CW);
  if (!pl.ok) return false;
  PlanCache pc;
  pc.plan = pl;
  if (hipMalloc((void**)&pc.d_frames, sizeof(FrameInfo) * pl.frames.size()) != hipSuccess) return false;
  if (hipMemcpy(pc.d_frames, pl.frames.data(), sizeof(FrameInfo) * pl.frames.size(), hipMemcpyHostToDevice) !=
      hipSuccess) {
    (void)hipFree(pc.d_frames);
    return false;
  }
  plan_table().push_back({p, pc});
  return true;
}

template <typename T, int S, int B, int REGK, int R>
static int launch_fused(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g,
                        const T* wts, const PlanCache& pc, double* partials, int* nblocks, bool final_reduce,
                        hipStream_t st) {
  using C = TileCfg<T, S>;
  constexpr int NP = (REGK == 2 ? 2 * R + 1 : 1);
  const HostPlan& pl = pc.plan;
  FusedArgs<T, B, NP> A;
  A.x = x; A.y = (const T*)p->d_obs; A.w = wts; A.g = g; A.partials = partials;
  A.counter = final_reduce ? (unsigned*)(p->d_cost + 4) : nullptr;  // d_cost[4..] is zero-initialised scratch
  A.cost_out = p->d_cost;
  A.frames = pc.d_frames;
  A.W = geo.W; A.H = geo.H; A.wl = geo.w; A.hl = geo.h; A.K = geo.K;
  A.obs_C = p->geo.C; A.obs_c0 = obs_c0;
  A.hu = pl.hu; A.hlc = pl.hlc;
  A.xrows = C::TH + pl.hu + pl.hd;
  A.xcells = C::CW + pl.hlc + pl.hrc;
  A.i0 = pl.i0; A.j0 = pl.j0; A.lrh = pl.lrh; A.lrw = pl.lrw;
  A.margin = pl.margin;
  A.terms = (int)terms;
  if (const char* dbg = getenv("SRMAP_DEBUG_SKIP")) A.terms |= atoi(dbg) << 8;  // ablation aid (profiling only)
  for (int i = 0; i < B * B; ++i) A.blur[i] = (T)p->blur2d[i];
  A.lambda = T(0);
  for (int i = 0; i < NP; ++i) A.powtab[i] = T(1);
  if (REGK != 0) {
    const RegSpec& rs = p->reg[pl.reg_index];
    A.lambda = (T)rs.lambda;
    if (REGK == 2) for (int i = 0; i < NP; ++i) A.powtab[i] = (T)rs.pow_table[i];
  }
  dim3 grid((geo.w + C::CW - 1) / C::CW, (geo.h + C::CH - 1) / C::CH, geo.C);
  hipLaunchKernelGGL((k_eval_fused<T, S, B, REGK, R>), grid, dim3(kThreads), 0, st, A);
  *nblocks = (int)(grid.x * grid.y * grid.z);
  SRMAP_HIP(p->ctx, hipGetLastError());
  return SRMAP_OK;
}

template <typename T, int S, int B>
static int dispatch_reg(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g,
                        const T* wts, const PlanCache& pc, int regk, int regr, double* partials, int* nb,
                        bool fr, hipStream_t st) {
  if (regk == 1) return launch_fused<T, S, B, 1, 0>(p, geo, obs_c0, terms, x, g, wts, pc, partials, nb, fr, st);
  if (regk == 2 && regr == 1) return launch_fused<T, S, B, 2, 1>(p, geo, obs_c0, terms, x, g, wts, pc, partials, nb, fr, st);
  if (regk == 2 && regr == 2) return launch_fused<T, S, B, 2, 2>(p, geo, obs_c0, terms, x, g, wts, pc, partials, nb, fr, st);
  if (regk == 2 && regr == 3) return launch_fused<T, S, B, 2, 3>(p, geo, obs_c0, terms, x, g, wts, pc, partials, nb, fr, st);
  return launch_fused<T, S, B, 0, 0>(p, geo, obs_c0, terms, x, g, wts, pc, partials, nb, fr, st);
}

template <typename T>
int launch_eval_tiled(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g,
                      double* partials, int* nblocks, hipStream_t st) {
  PlanCache* pc = find_plan(p);
  if (!pc) return set_error(p->ctx, SRMAP_EUNSUPPORTED, "no tile plan");
  const HostPlan& pl = pc->plan;
  const size_t N = (size_t)geo.W * geo.H;
  // which regularisers the fused kernel takes, which go to the direct kernels
  int regk = 0, regr = 0;
  const bool want_reg = (terms & SRMAP_TERM_REG) != 0;
  const T* wts = nullptr;
  if (want_reg && pl.regk != 0 && !(pl.regk == 2 && geo.s == 2 && pl.regr == 3 && false)) {
    regk = pl.regk; regr = pl.regr;
    const RegSpec& rs = p->reg[pl.reg_index];
    wts = rs.weights ? (const T*)rs.weights + (size_t)obs_c0 * N : nullptr;
  }
  unsigned fused_terms = terms & SRMAP_TERM_DATA;
  if (regk) fused_terms |= SRMAP_TERM_REG;
  // the fused kernel finishes the cost reduction itself when it is the only
  // producer of partials
  bool extra = false;
  if (want_reg)
    for (int r = 0; r < p->nreg; ++r)
      if (!(regk && r == pl.reg_index) && p->reg[r].lambda > 0.0) extra = true;
  const bool fr = !extra;
  int rc = SRMAP_OK, nb = 0;
  if (geo.s == 2 && geo.b == 1) rc = dispatch_reg<T, 2, 1>(p, geo, obs_c0, fused_terms, x, g, wts, *pc, regk, regr, partials, &nb, fr, st);
  else if (geo.s == 2 && geo.b == 3) rc = dispatch_reg<T, 2, 3>(p, geo, obs_c0, fused_terms, x, g, wts, *pc, regk, regr, partials, &nb, fr, st);
  else if (geo.s == 3 && geo.b == 1) rc = dispatch_reg<T, 3, 1>(p, geo, obs_c0, fused_terms, x, g, wts, *pc, regk, regr, partials, &nb, fr, st);
  else if (geo.s == 3 && geo.b == 3) rc = dispatch_reg<T, 3, 3>(p, geo, obs_c0, fused_terms, x, g, wts, *pc, regk, regr, partials, &nb, fr, st);
  else if (geo.s == 4 && geo.b == 1) rc = dispatch_reg<T, 4, 1>(p, geo, obs_c0, fused_terms, x, g, wts, *pc, regk, regr, partials, &nb, fr, st);
  else if (geo.s == 4 && geo.b == 3) rc = dispatch_reg<T, 4, 3>(p, geo, obs_c0, fused_terms, x, g, wts, *pc, regk, regr, partials, &nb, fr, st);
  else return set_error(p->ctx, SRMAP_EUNSUPPORTED, "no fused kernel for scale %d blur %d", geo.s, geo.b);
  if (rc) return rc;
  if (fr) { *nblocks = 0; return SRMAP_OK; }  // total already in d_cost[0]
  int total = nb;
  // remaining regularisers (TV3D, a second regulariser, BTV range > 3): direct kernels
  if (want_reg) {
    for (int r = 0; r < p->nreg; ++r) {
      if (regk && r == pl.reg_index) continue;
      const RegSpec& rs = p->reg[r];
      if (rs.lambda <= 0.0) continue;
      if (!p->d_regvals) SRMAP_HIP(p->ctx, hipMalloc(&p->d_regvals, p->hr_count() * sizeof(T)));
      rc = launch_reg_values<T>(p, geo, rs, x, (T*)p->d_regvals, st);
      if (rc) return rc;
      const T* w2 = rs.weights ? (const T*)rs.weights + (size_t)obs_c0 * N : nullptr;
      int nb2 = 0;
      rc = launch_reg_gradient_direct<T>(p, geo, rs, x, w2, rs.lambda, (const T*)p->d_regvals, g, true,
                                         partials + total, &nb2, st);
      if (rc) return rc;
      total += nb2;
    }
  }
  *nblocks = total;
  return SRMAP_OK;
}

template int launch_eval_tiled<float>(srmap_problem*, const Geometry&, int, unsigned, const float*, float*,
                                      double*, int*, hipStream_t);
template int launch_eval_tiled<double>(srmap_problem*, const Geometry&, int, unsigned, const double*,
                                       double*, double*, int*, hipStream_t);

}  // namespace srmap
