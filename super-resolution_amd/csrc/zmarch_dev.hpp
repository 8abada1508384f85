// zmarch_dev.hpp -- device helpers of the MARCHING evaluation kernel k_eval_m (kernels_zmarch.hip).
//
// Same formulation as the tile kernel (ztile_dev.hpp: owner computes on the HR grid, objective_function.cpp:5-20), other
// decomposition: ONE resident workgroup of 16 waves per CU owns a STRIP of 64 LR cells (64 S pixels) x a BAND of rows and
// walks down it 16 rows per step.  x rows, zh and 2*lambda*w*r live in LDS as RINGS of rows; the rows a step adds to the
// x window are requested one step ahead by direct-to-LDS loads (global_load_lds_dwordx4: no staging registers, no
// ds_write), in a layout of 16-byte GRANULES [plane][cell] that every window reader fetches with ds_read_b128.
//   * x is staged RAW (the request cannot scale it): the 2^Q factor of the sign clamps (ztile_dev.hpp, Pre<T>) moves into
//     the difference itself, d' = fma(-2^Q, x[q], x[p] * 2^Q) -- one instruction like the subtraction it replaces and
//     bit-identical to (x[p] - x[q]) * 2^Q of the pre-scaled tile (scaling by a power of two commutes with rounding).
//   * halo rows of zh / 2*lambda*w*r are evaluated once per BAND (a "virtual step" in front of the first one), not
//     per 8 rows; a workgroup's launch, argument fetch and address arithmetic are paid once per band.
#pragma once
#include "ztile_dev.hpp"

namespace srmap {
namespace {

template <typename T, int S, int B, int REGK, int R>
struct MCfg {
  using Z = ZCfg<T, S, B, REGK, R>;
  static constexpr int NW = 16;               // waves = HR rows per step
  static constexpr int NT = 64 * NW;
  static constexpr int SR = NW;
  static constexpr int CW = 64;
  static constexpr int TW = CW * S;
  static constexpr int HB = Z::HB, WIN = Z::WIN, RU = Z::RU, NV = Z::NV, NP = Z::NP;
  static constexpr int G = 16 / (int)sizeof(T);   // pixels per granule
  static constexpr int PL = S / G;                // granules (planes) per cell; S % G == 0 is a condition of the plan
  static constexpr int ZA = (B > 1) ? 1 : 0;      // zh is evaluated ZA rows ahead of the gradient rows
  static constexpr int XCL = Z::XCL, XCR = Z::XCR, XC = Z::XC;
  static constexpr int XRG = PL * XC;             // granules per x row
  static constexpr int NLD = (XRG + 63) / 64;     // direct-to-LDS requests per x row
  static constexpr int XLO = zmax(RU, (B > 1) ? 2 * HB : 0);   // x rows above the step's first row
  static constexpr int XHI = zmax(WIN, ZA + HB);               // x rows below the step's last row
  static constexpr int XWIN = XLO + SR + XHI;     // x rows a step reads
  static constexpr int NXR = XWIN + SR;           // ring: window + the next step's SR new rows
  static constexpr int ZRG = PL * CW;
  static constexpr int NZR = (B > 1) ? SR + 2 * HB : 1;
  static constexpr int CCL = Z::CCL, CC = Z::CC;
  static constexpr int CRG = PL * CC;
  static constexpr int NCR = REGK ? SR + RU : 1;
  static constexpr int VW = zmax(RU, (B > 1) ? 2 : 0);  // waves of the virtual step: SR - VW .. SR - 1
  // pixel windows (relative to the first pixel of the thread's cell)
  static constexpr int P1LO = (B > 1) ? -2 * HB : 0, P1HI = zmax(S - 1 + WIN, S - 1 + 2 * HB);
  static constexpr int P2LO = -RU, P2HI = S - 1;
};

template <typename T, int G> struct Gran { typedef T type __attribute__((ext_vector_type(G))); };

// Pixels [PLO, PHI] of one LDS row in granule layout [plane][ROWC cells] with LEFTC halo cells on the left; `base` =
// first granule of the row + lane.  Every granule is one ds_read_b128 at an immediate offset.
template <typename T, int S, int ROWC, int LEFTC, int PLO, int PHI>
struct PWin {
  static constexpr int G = 16 / (int)sizeof(T);
  using GT = typename Gran<T, G>::type;
  static constexpr int GLO = floordiv(PLO, G), GHI = floordiv(PHI, G), NG = GHI - GLO + 1;
  GT g[NG];
  static __device__ __forceinline__ constexpr int gidx(int gn) {  // granule number gn = floor(pixel / G)
    return (posmod(gn * G, S) / G) * ROWC + LEFTC + floordiv(gn * G, S);
  }
  __device__ __forceinline__ void load(const GT* __restrict__ base) {
#pragma unroll
    for (int i = 0; i < NG; ++i) g[i] = base[gidx(GLO + i)];
  }
  __device__ __forceinline__ T at(int p) const { return g[floordiv(p, G) - GLO][posmod(p, G)]; }
};

template <int N>
__device__ __forceinline__ int mwrap(int v) { return v >= N ? v - N : v; }   // v in [0, 2N)
template <int N>
__device__ __forceinline__ int mwrapn(int v) { return v < 0 ? v + N : v; }   // v in [-N, N)

// sgn / step of a pre-scaled difference, as in ztile_dev.hpp
template <typename T> __device__ __forceinline__ T m_scale() { return Pre<T>::up(T(1)); }

// ---- phase 1 of one wave: regulariser pass 1 for HR row gr and data term (B x, residuals, z, horizontal half of B^T)
// for HR row zrow = gr + ZA, from the SAME window rows (x rows gr .. gr + WIN).
//   xb[i]   first granule (+ lane) of x row gr + i
//   zdst    first granule (+ lane) of the zh row's slot,  cdst: of the 2*lambda*w*r row's slot
//   do_z / do_r   this wave evaluates the data / regulariser part (uniform)
//   count_z       the residual row belongs to this band (its cost is counted here)
//   full          regulariser row belongs to this band (self term, cost); else only 2*lambda*w*r (halo rows)
// SLOW: rows / strips at the image border -- LR validity masks, dropped blur taps of LR row / column 0, window taps
// outside the image (the EDGE / BORDER paths of ztile_dev.hpp).
template <typename T, int S, int B, int REGK, int R, bool SLOW, typename ArgsT>
__device__ __forceinline__ void m_phase1(const ArgsT& A, const typename Gran<T, 16 / (int)sizeof(T)>::type* const (&xb)[MCfg<T, S, B, REGK, R>::WIN + 1],
                                         typename Gran<T, 16 / (int)sizeof(T)>::type* zdst,
                                         typename Gran<T, 16 / (int)sizeof(T)>::type* cdst, bool do_z, bool do_r, bool count_z,
                                         bool full, bool cost_row, int gr, int zrow, int CJ0, int lane,
                                         const T* __restrict__ ybase, const T (&ypre)[MCfg<T, S, B, REGK, R>::NV],
                                         const T (&wv)[S], T (&acc)[S], T (&zown)[S], double& cost_data, double& cost_reg) {
  using C = MCfg<T, S, B, REGK, R>;
  using ZC = typename C::Z;
  using GT = typename Gran<T, C::G>::type;
  constexpr int HB = C::HB, NV = C::NV, WIN = C::WIN, G = C::G, PL = C::PL;
  const int gc0 = (CJ0 + lane) * S;
  const T SC = m_scale<T>();
  T bx[NV], btop[NV], bleft[NV], bcorner[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) { bx[v] = T(0); btop[v] = T(0); bleft[v] = T(0); bcorner[v] = T(0); }
  T x0s[S], rv[S], dv[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) { rv[pc] = T(0); dv[pc] = T(0); x0s[pc] = T(0); }
  constexpr int NROW = zmax(WIN + 1, B);
#pragma unroll
  for (int i = 0; i < NROW; ++i) {
    PWin<T, S, C::XC, C::XCL, C::P1LO, C::P1HI> w;
    w.load(xb[i < WIN + 1 ? i : WIN]);
    if (i < B && do_z) {  // blur row a = i of the residual row zrow (x row zrow - HB + a = gr + i when ZA == HB)
      constexpr int dummy = 0; (void)dummy;
      const int a = i;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
#pragma unroll
        for (int e = 0; e < B; ++e) bx[v] += blur_tap<B>(A, a, e) * w.at(v + e - 2 * HB);
        if (SLOW && B > 1) {
          bleft[v] += blur_tap<B>(A, a, 0) * w.at(v - 2 * HB);
          if (a == 0) {
#pragma unroll
            for (int e = 0; e < B; ++e) btop[v] += blur_tap<B>(A, 0, e) * w.at(v + e - 2 * HB);
            bcorner[v] = blur_tap<B>(A, 0, 0) * w.at(v - 2 * HB);
          }
        }
      }
    }
    if (REGK != 0 && i <= WIN && do_r) {
      if (i == 0) {
#pragma unroll
        for (int pc = 0; pc < S; ++pc) x0s[pc] = w.at(pc) * SC;
      }
#pragma unroll
      for (int pc = 0; pc < S; ++pc) {
        if (REGK == 2) {
#pragma unroll
          for (int j = 0; j <= R; ++j) {
            if (i == 0 && j == 0) continue;
            T d = __builtin_fma(-SC, w.at(pc + j), x0s[pc]);   // (x[p] - x[q]) * 2^Q, one rounding
            if (SLOW) d = ((gr + i < A.H) && (gc0 + pc + j < A.W)) ? d : T(0);
            rv[pc] += A.powtab[i + j] * absv(d);
            if (i < R && j < R) {
              if (sizeof(T) == 8) dv[pc] += A.powtab[i + j] * step_pre<T>(d);
              else dv[pc] += sgn_pre<T>(d, A.powtab[i + j]);
            }
          }
        } else if (i == 1) {
          T dyv = __builtin_fma(SC, w.at(pc), -x0s[pc]);
          if (SLOW) dyv = (gr + 1 < A.H) ? dyv : T(0);
          rv[pc] = absv(dyv) + rv[pc];
          dv[pc] = dv[pc] - sgn_pre<T>(dyv, T(1));
        } else if (i == 0) {
          T dxv = __builtin_fma(SC, w.at(pc + 1), -x0s[pc]);
          if (SLOW) dxv = (gc0 + pc + 1 < A.W) ? dxv : T(0);
          rv[pc] = absv(dxv);
          dv[pc] = -sgn_pre<T>(dxv, T(1));
        }
      }
    }
  }
  // ---- regulariser: 2*lambda*w*r, self term, cost (tv_regularizer.cpp:110-170, btv_regularizer.cpp:19-136) ----
  if (REGK != 0 && do_r) {
    T cr2v[S];
#pragma unroll
    for (int pc = 0; pc < S; ++pc) {
      const T r = Pre<T>::down(rv[pc]);
      const T c = A.lambda * wv[pc];
      T cr2 = T(2) * c * r;
      const bool in_img = (unsigned)gr < (unsigned)A.H && (unsigned)(gc0 + pc) < (unsigned)A.W;
      if (REGK == 2 && sizeof(T) == 8) dv[pc] = T(2) * dv[pc] - A.pwsum;
      const T selfv = cr2 * dv[pc];
      acc[pc] += full ? selfv : T(0);
      const double cd = (in_img && cost_row && full) ? (double)c * (double)r * (double)r : 0.0;
      cost_reg += cd;
      if (!in_img || (REGK == 2 && gr == 0 && gc0 + pc == 0)) cr2 = T(0);
      cr2v[pc] = cr2;
    }
#pragma unroll
    for (int pl = 0; pl < PL; ++pl) {
      GT o;
#pragma unroll
      for (int e = 0; e < G; ++e) o[e] = cr2v[pl * G + e];
      cdst[pl * C::CC + C::CCL] = o;
    }
  }
  // ---- data term: residuals of the frames whose LR grid hits each pixel (objective_data_term.cpp:15-75) ----
  if (do_z) {
    int rc, pr;
    row_phase<S>(zrow, rc, pr);
    int cn[S];
#pragma unroll
    for (int pc = 0; pc < S; ++pc) cn[pc] = A.cntk[pr][pc];
    const int mmax = A.cntk[pr][S];
    const int mfull = SLOW ? 0 : A.cntk[pr][S + 1];
    T z[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) z[v] = T(0);
    for (int t = 0; t < mmax; ++t) {
      T yv[NV];
      if (t == 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) yv[v] = ypre[v];
      } else {
        load_obs_row<T, S, ZC, SLOW>(A, pr, rc, t, CJ0, lane, ybase, cn, yv);
      }
      if (!SLOW && t < mfull) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int pcv = v - HB;
          const T rr = bx[v] - yv[v];
          z[v] += rr;
          if (pcv >= 0 && pcv < S && count_z) cost_data += (double)rr * (double)rr;
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
          if (!SLOW) {
            rr = bx[v] - yv[v];
            z[v] += rr;
            if (own && count_z) cost_data += (double)rr * (double)rr;
          } else {
            const ZEntry e = (t == 0) ? aux0_at(A, pr, pc) : ctab(A.aux, slot + pc);
            const int i = rc + e.io, j = CJ0 + lane + dc + e.jo;
            T bxv = bx[v];
            if (B > 1) {
              // filter2D's zero padding acts on the warped image: LR row 0 loses blur tap row 0, LR column 0 tap column 0
              const bool j0 = j == 0;
              if (i == 0) bxv = bxv - btop[v] - (j0 ? bleft[v] - bcorner[v] : T(0));
              else bxv = bxv - (j0 ? bleft[v] : T(0));
            }
            rr = bxv - yv[v];
            rr = ((unsigned)i < (unsigned)A.hl && (unsigned)j < (unsigned)A.wl) ? rr : T(0);  // no such LR pixel
            z[v] += rr;
            if (own && count_z && S * i >= A.cr0 && S * i < A.cr1) {
              const bool in_img = zrow < A.H && gc0 + (own ? pcv : 0) < A.W;
              const double rd = (double)(in_img ? rr : T(0));
              cost_data += rd * (double)rr;
            }
          }
        }
      }
    }
    if (B == 1) {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) zown[pc] = z[pc];
    } else {
      T zh[S];
#pragma unroll
      for (int pc = 0; pc < S; ++pc) {
        zh[pc] = T(0);
#pragma unroll
        for (int e = 0; e < B; ++e) zh[pc] += k1_tap<B>(A, e) * z[pc + e];
      }
#pragma unroll
      for (int pl = 0; pl < PL; ++pl) {
        GT o;
#pragma unroll
        for (int e = 0; e < G; ++e) o[e] = zh[pl * G + e];
        zdst[pl * C::CW] = o;
      }
    }
  }
}

// 2*lambda*w*r of ONE left-halo-column pixel (column COL < 0 relative to the strip), one row per lane.
//   xe[i]  per-lane ELEMENT offset of x row gr + i's slot,  ce: of the 2*lambda*w*r row's slot (xs / cs as T arrays)
template <typename T, int S, int B, int REGK, int R, int COL, bool BORDER, typename ArgsT>
__device__ __forceinline__ void m_halo_col(const ArgsT& A, const T* __restrict__ xs, T* __restrict__ cs, T wt,
                                           const int (&xe)[MCfg<T, S, B, REGK, R>::WIN + 1], int ce, int gr, int gc) {
  using C = MCfg<T, S, B, REGK, R>;
  constexpr int WIN = C::WIN, G = C::G;
  auto xel = [](int p) constexpr { return ((posmod(floordiv(p, G) * G, S) / G) * C::XC + C::XCL + floordiv(floordiv(p, G) * G, S)) * G + posmod(p, G); };
  auto cel = [](int p) constexpr { return ((posmod(floordiv(p, G) * G, S) / G) * C::CC + C::CCL + floordiv(floordiv(p, G) * G, S)) * G + posmod(p, G); };
  const T SC = m_scale<T>();
  T cr2 = T(0);
  if (gr >= 0 && gr < A.H && gc >= 0 && gc < A.W && !(REGK == 2 && gr == 0 && gc == 0)) {
    const T x0 = xs[xe[0] + xel(COL)] * SC;
    T r = T(0);
    if (REGK == 2) {
#pragma unroll
      for (int i = 0; i <= WIN; ++i) {
#pragma unroll
        for (int j = 0; j <= WIN; ++j) {
          if (i == 0 && j == 0) continue;
          const T v = xs[xe[i] + xel(COL + j)];
          const T d = (!BORDER || (gr + i < A.H && gc + j < A.W)) ? __builtin_fma(-SC, v, x0) : T(0);
          r += A.powtab[i + j] * absv(d);
        }
      }
    } else {
      const T yv = (!BORDER || gr + 1 < A.H) ? absv(__builtin_fma(SC, xs[xe[1] + xel(COL)], -x0)) : T(0);
      const T xv = (!BORDER || gc + 1 < A.W) ? absv(__builtin_fma(SC, xs[xe[0] + xel(COL + 1)], -x0)) : T(0);
      r = yv + xv;
    }
    cr2 = T(2) * (A.lambda * wt) * Pre<T>::down(r);
  }
  cs[ce + cel(COL)] = cr2;
}

// ---- phase 2 of one wave (HR row gr): vertical half of B^T, regulariser pass 2 (tv_regularizer.cpp:172-203,
// btv_regularizer.cpp:137-162); acc holds the self term of pass 1 on entry, the gradient on exit ----
//   xb[i] / cb[i]  first granule (+ lane) of x / 2*lambda*w*r row gr - i;  zb[a]: of zh row gr - HB + a
template <typename T, int S, int B, int REGK, int R, typename ArgsT>
__device__ __forceinline__ void m_phase2(const ArgsT& A, const typename Gran<T, 16 / (int)sizeof(T)>::type* const (&xb)[MCfg<T, S, B, REGK, R>::RU + 1],
                                         const typename Gran<T, 16 / (int)sizeof(T)>::type* const (&cb)[MCfg<T, S, B, REGK, R>::RU + 1],
                                         const typename Gran<T, 16 / (int)sizeof(T)>::type* const (&zb)[B], bool want_data, bool want_reg,
                                         const T (&zown)[S], T (&acc)[S]) {
  using C = MCfg<T, S, B, REGK, R>;
  constexpr int RU = C::RU, G = C::G;
  const T SC = m_scale<T>();
  if (want_data) {
    const T sc = (T)(2 * S * S);  // g += 2 * (s*s block sum) (objective_data_term.cpp:55-71)
    T zz[S];
    if (B == 1) {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) zz[pc] = zown[pc];
    } else {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) zz[pc] = T(0);
#pragma unroll
      for (int a = 0; a < B; ++a) {
        PWin<T, S, C::CW, 0, 0, S - 1> w;
        w.load(zb[a]);
#pragma unroll
        for (int pc = 0; pc < S; ++pc) zz[pc] += k1_tap<B>(A, a) * w.at(pc);
      }
    }
#pragma unroll
    for (int pc = 0; pc < S; ++pc) acc[pc] += sc * zz[pc];
  }
  if (REGK != 0 && RU > 0 && want_reg) {
    T x0s[S], sum[S];
#pragma unroll
    for (int pc = 0; pc < S; ++pc) sum[pc] = T(0);
#pragma unroll
    for (int i = 0; i <= RU; ++i) {
      PWin<T, S, C::XC, C::XCL, C::P2LO, C::P2HI> xw;
      PWin<T, S, C::CC, C::CCL, C::P2LO, C::P2HI> cw;
      xw.load(xb[i]);
      cw.load(cb[i]);
      if (i == 0) {
#pragma unroll
        for (int pc = 0; pc < S; ++pc) x0s[pc] = xw.at(pc) * SC;
      }
#pragma unroll
      for (int pc = 0; pc < S; ++pc) {
        if (REGK == 2) {
          if (i < R) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
              if (i == 0 && j == 0) continue;
              // -sgn(x[q] - x[p]) * alpha^(i+j) * 2 c[q] r[q],  q = p - (i, j)
              sum[pc] += cw.at(pc - j) * sgn_pre<T>(__builtin_fma(-SC, xw.at(pc - j), x0s[pc]), A.powtab[i + j]);
            }
          }
        } else {
          if (i == 0) sum[pc] += cw.at(pc - 1) * sgn_pre<T>(__builtin_fma(-SC, xw.at(pc - 1), x0s[pc]), T(1));
          else sum[pc] += cw.at(pc) * sgn_pre<T>(__builtin_fma(-SC, xw.at(pc), x0s[pc]), T(1));
        }
      }
    }
#pragma unroll
    for (int pc = 0; pc < S; ++pc) acc[pc] += sum[pc];
  }
}

}  // namespace
}  // namespace srmap
