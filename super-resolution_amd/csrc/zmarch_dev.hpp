// zmarch_dev.hpp -- device helpers of the MARCHING tile kernel (k_eval_m in kernels_ztile.hip).
//
// A workgroup of 8 waves owns a BAND of 8 * nsteps HR rows x 64 * S columns and walks down it eight rows at a time.  The
// x tile, zh and 2*lambda*w*r live in LDS as RINGS of rows (13 / 10 / 10 slots, the tile kernel's footprint): a step
// replaces the eight oldest rows, so x rows are loaded once per band instead of 13 per 8, zh / 2*lambda*w*r halo rows are
// evaluated once per band instead of once per tile, and a workgroup's launch, argument fetch and input latency are
// paid once per band.  The helpers below are the tile kernel's (ztile_dev.hpp) with every LDS row addressed through a
// wave-uniform ROW BASE (element offset of the row's ring slot) instead of `row * ROWSTRIDE`.
#pragma once
#include "ztile_dev.hpp"

#ifndef SRMAP_EXP_MSB
#define SRMAP_EXP_MSB 0   // 1: scheduling barriers between the window rows of a pass (caps the LDS read-ahead, i.e. the registers)
#endif
#define M_SCHED_ROW do { if (SRMAP_EXP_MSB) __builtin_amdgcn_sched_barrier(0); } while (0)

namespace srmap {
namespace {

// column part of an x-tile / 2*lambda*w*r index (pixel column relative to the first pixel of the thread's cell)
template <typename C>
__device__ __forceinline__ constexpr int xcol(int col) { return xi<C>(0, col); }
template <typename C>
__device__ __forceinline__ constexpr int ccol(int col) { return ci<C>(0, col); }

// ---- data term, phase 1 (z_row of ztile_dev.hpp; objective_data_term.cpp:15-75 per HR pixel) ----
// grow: global HR row of the residuals' z positions; xb[a]: ring base of x row grow + a - HB; zb: ring base of the zh row.
template <typename T, int S, int B, typename C, bool EDGE, typename ArgsT>
__device__ __forceinline__ void z_row_m(const ArgsT& A, const T* __restrict__ xs, T* __restrict__ zs, int grow,
                                        const int (&xb)[B], int zb, int cell0, int lane, const T* __restrict__ ybase,
                                        const T (&ypre)[C::NV], bool count, const T (&mk)[S], T (&zout)[S], double& cost) {
  constexpr int HB = C::HB, NV = C::NV;
  int rc, pr;
  row_phase<S>(grow, rc, pr);
  T bx[NV], btop[NV], bleft[NV], bcorner[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) { bx[v] = T(0); btop[v] = T(0); bleft[v] = T(0); bcorner[v] = T(0); }
#pragma unroll
  for (int a = 0; a < B; ++a) {
    T xr[NV + B - 1];
#pragma unroll
    for (int j = 0; j < NV + B - 1; ++j) xr[j] = xs[xb[a] + xcol<C>(j - 2 * HB) + lane];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
      for (int e = 0; e < B; ++e) bx[v] += blur_tap<B>(A, a, e) * xr[v + e];
      if (EDGE && B > 1) {
        bleft[v] += blur_tap<B>(A, a, 0) * xr[v];  // tap column 0
        if (a == 0) {
#pragma unroll
          for (int e = 0; e < B; ++e) btop[v] += blur_tap<B>(A, 0, e) * xr[v + e];  // tap row 0
          bcorner[v] = blur_tap<B>(A, 0, 0) * xr[v];
        }
      }
    }
    M_SCHED_ROW;
  }
  const T unscale = Pre<T>::down(T(1));
  int cn[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) cn[pc] = A.cntk[pr][pc];
  const int mmax = A.cntk[pr][S];
  const int mfull = EDGE ? 0 : A.cntk[pr][S + 1];
  T z[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) z[v] = T(0);
  for (int t = 0; t < mmax; ++t) {
    T yv[NV];
    if (t == 0) {
#pragma unroll
      for (int v = 0; v < NV; ++v) yv[v] = ypre[v];
    } else {
      load_obs_row<T, S, C, EDGE>(A, pr, rc, t, cell0, lane, ybase, cn, yv);
    }
    if (!EDGE && t < mfull) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int pcv = v - HB;
        const T rr = bx[v] * unscale - yv[v];
        z[v] += rr;
        if (pcv >= 0 && pcv < S && count) cost += (double)rr * (double)rr;
      }
      continue;
    }
    const size_t slot = (size_t)(t * S + pr) * S;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int pcv = v - HB, pc = posmod(pcv, S), dc = floordiv(pcv, S);
      const bool own = pcv >= 0 && pcv < S;
      if (t < cn[pc]) {  // uniform
        T rr;
        if (!EDGE) {
          rr = bx[v] * unscale - yv[v];
          z[v] += rr;
          if (own && count) cost += (double)rr * (double)rr;
        } else {
          const ZEntry e = (t == 0) ? aux0_at(A, pr, pc) : ctab(A.aux, slot + pc);
          const int i = rc + e.io, j = cell0 + lane + dc + e.jo;
          T bxv = bx[v];
          if (B > 1) {
            // filter2D's zero padding acts on the warped image: LR row 0 loses blur tap row 0, LR column 0 tap column 0
            const bool j0 = j == 0;
            if (i == 0) bxv = bxv - btop[v] - (j0 ? bleft[v] - bcorner[v] : T(0));
            else bxv = bxv - (j0 ? bleft[v] : T(0));
          }
          rr = bxv * unscale - yv[v];
          rr = ((unsigned)i < (unsigned)A.hl && (unsigned)j < (unsigned)A.wl) ? rr : T(0);  // no such LR pixel
          z[v] += rr;
          if (own && count && S * i >= A.cr0 && S * i < A.cr1) {
            const double rd = (double)(rr * mk[own ? pcv : 0]);
            cost += rd * (double)rr;
          }
        }
      }
    }
  }
  if (B == 1) {
#pragma unroll
    for (int pc = 0; pc < S; ++pc) zout[pc] = z[pc];
  } else {
#pragma unroll
    for (int pc = 0; pc < S; ++pc) {
      T zh = T(0);
#pragma unroll
      for (int e = 0; e < B; ++e) zh += k1_tap<B>(A, e) * z[pc + e];
      zs[zb + pc * C::CW + lane] = zh;
    }
  }
}

// t = 0 observations of the NV pixels of the thread's cell in global HR row grow
template <typename T, int S, int B, typename C, typename ArgsT>
__device__ __forceinline__ void z_prefetch_m(const ArgsT& A, int grow, int cell0, int lane, bool edge,
                                             const T* __restrict__ ybase, T (&ypre)[C::NV]) {
  int rc, pr;
  row_phase<S>(grow, rc, pr);
  int cn[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) cn[pc] = A.cntk[pr][pc];
  if (edge) load_obs_row<T, S, C, true>(A, pr, rc, 0, cell0, lane, ybase, cn, ypre);
  else load_obs_row<T, S, C, false>(A, pr, rc, 0, cell0, lane, ybase, cn, ypre);
}

// ---- regulariser pass 1 (reg_row of ztile_dev.hpp; tv_regularizer.cpp:110-170, btv_regularizer.cpp:19-136) ----
// xb[i]: ring base of x row gr + i (i = 0 .. WIN); cb: ring base of the 2*lambda*w*r row of gr.
template <typename T, int S, int REGK, int R, typename C, bool BORDER, bool FULL>
__device__ __forceinline__ void reg_row_m(T (&acc)[S], double& cost, const T* __restrict__ xs, T* __restrict__ cs,
                                          const T (&wv)[S], const int (&xb)[C::WIN + 1], int cb, int lane, int gr, int gc0,
                                          int W, int H, T lambda, const T (&pw)[C::NP], T pwsum, bool cost_row) {
  constexpr int WIN = C::WIN;
  constexpr int NC = S + WIN;
  T x0v[S], rv[S], dv[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) { rv[pc] = T(0); dv[pc] = T(0); }
#pragma unroll
  for (int i = 0; i <= WIN; ++i) {
    T row[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) row[j] = xs[xb[i] + xcol<C>(j) + lane];
    if (i == 0) {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) x0v[pc] = row[pc];
    }
#pragma unroll
    for (int pc = 0; pc < S; ++pc) {
      if (REGK == 2) {
#pragma unroll
        for (int j = 0; j <= R; ++j) {
          if (i == 0 && j == 0) continue;
          T d = x0v[pc] - row[pc + j];
          if (BORDER) d = ((gr + i < H) && (gc0 + pc + j < W)) ? d : T(0);
          rv[pc] += pw[i + j] * absv(d);
          if (FULL && i < R && j < R) {
            if (sizeof(T) == 8) dv[pc] += pw[i + j] * step_pre<T>(d);
            else dv[pc] += sgn_pre<T>(d, pw[i + j]);
          }
        }
      } else if (i == 1) {
        T dyv = row[pc] - x0v[pc];
        if (BORDER) dyv = (gr + 1 < H) ? dyv : T(0);
        rv[pc] = absv(dyv) + rv[pc];
        if (FULL) dv[pc] = dv[pc] - sgn_pre<T>(dyv, T(1));
      } else {
        T dxv = row[pc + 1] - x0v[pc];
        if (BORDER) dxv = (gc0 + pc + 1 < W) ? dxv : T(0);
        rv[pc] = absv(dxv);
        if (FULL) dv[pc] = -sgn_pre<T>(dxv, T(1));
      }
    }
    M_SCHED_ROW;
  }
#pragma unroll
  for (int pc = 0; pc < S; ++pc) {
    const T r = Pre<T>::down(rv[pc]);
    const T c = lambda * wv[pc];
    T cr2 = T(2) * c * r;
    const bool in_img = (unsigned)gr < (unsigned)H && (unsigned)(gc0 + pc) < (unsigned)W;
    if (FULL) {
      if (REGK == 2 && sizeof(T) == 8) dv[pc] = T(2) * dv[pc] - pwsum;
      acc[pc] += cr2 * dv[pc];
      const double cd = (in_img && cost_row) ? (double)c * (double)r * (double)r : 0.0;
      cost += cd;
    }
    if (!in_img || (REGK == 2 && gr == 0 && gc0 + pc == 0)) cr2 = T(0);
    cs[cb + ccol<C>(pc) + lane] = cr2;
  }
}

// 2*lambda*w*r of ONE left-halo-column pixel, one row per lane.  xo[i]: per-lane element offset of x row gr + i's ring
// slot, co: per-lane element offset of the 2*lambda*w*r row's slot.
template <typename T, int S, int REGK, int R, typename C, int COL, bool BORDER>
__device__ __forceinline__ void reg_halo_col_m(const T* __restrict__ xs, T* __restrict__ cs, const T wt,
                                               const int (&xo)[C::WIN + 1], int co, int gr, int gc, int W, int H, T lambda,
                                               const T (&pw)[C::NP]) {
  constexpr int WIN = C::WIN;
  T cr2 = T(0);
  if (gr >= 0 && gr < H && gc >= 0 && gc < W && !(REGK == 2 && gr == 0 && gc == 0)) {
    const T x0 = xs[xo[0] + xcol<C>(COL)];
    T r = T(0);
    if (REGK == 2) {
#pragma unroll
      for (int i = 0; i <= WIN; ++i) {
#pragma unroll
        for (int j = 0; j <= WIN; ++j) {
          if (i == 0 && j == 0) continue;
          const T v = xs[xo[i] + xcol<C>(COL + j)];
          const T d = (!BORDER || (gr + i < H && gc + j < W)) ? x0 - v : T(0);
          r += pw[i + j] * absv(d);
        }
      }
    } else {
      const T yv = (!BORDER || gr + 1 < H) ? absv(xs[xo[1] + xcol<C>(COL)] - x0) : T(0);
      const T xv = (!BORDER || gc + 1 < W) ? absv(xs[xo[0] + xcol<C>(COL + 1)] - x0) : T(0);
      r = yv + xv;
    }
    cr2 = T(2) * (lambda * wt) * Pre<T>::down(r);
  }
  cs[co + ccol<C>(COL)] = cr2;
}

// ---- regulariser pass 2 (reg_pass2z of ztile_dev.hpp; tv_regularizer.cpp:172-203, btv_regularizer.cpp:137-162) ----
// xb[i] / cb[i]: ring bases of row r - i (i = 0 .. RU).
template <typename T, int S, int REGK, int R, typename C>
__device__ __forceinline__ void reg_pass2_m(T (&acc)[S], const T* __restrict__ xs, const T* __restrict__ cs,
                                            const int (&xb)[C::RU + 1], const int (&cb)[C::RU + 1], int lane,
                                            const T (&pw)[C::NP]) {
  constexpr int RU = C::RU;
  if (RU == 0) return;
  constexpr int NC = S + RU;
  T x0v[S], sum[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) sum[pc] = T(0);
#pragma unroll
  for (int i = 0; i <= RU; ++i) {
    T xw[NC], cw[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      xw[j] = xs[xb[i] + xcol<C>(j - RU) + lane];
      cw[j] = cs[cb[i] + ccol<C>(j - RU) + lane];
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
            sum[pc] += cw[pc + RU - j] * sgn_pre<T>(x0v[pc] - xw[pc + RU - j], pw[i + j]);
          }
        }
      } else {
        if (i == 0) sum[pc] += cw[pc + RU - 1] * sgn_pre<T>(x0v[pc] - xw[pc + RU - 1], T(1));
        else sum[pc] += cw[pc + RU] * sgn_pre<T>(x0v[pc] - xw[pc + RU], T(1));
      }
    }
    M_SCHED_ROW;
  }
#pragma unroll
  for (int pc = 0; pc < S; ++pc) acc[pc] += sum[pc];
}

// ring slot of tile-relative row rel (>= -LO) for a ring of N slots whose slot 0 holds row `base - LO`... in short:
// (b0 + rel + LO) mod N with b0 in [0, N) and rel + LO in [0, 2N)
template <int N>
__device__ __forceinline__ int ring_slot(int b0, int v) {
  int s = b0 + v;
  s = s >= N ? s - N : s;
  s = s >= N ? s - N : s;
  return s;
}

}  // namespace
}  // namespace srmap
