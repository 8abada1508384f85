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
//   * the image border costs no second code path: the plan admits frame offsets in [-(S-1), 0] only (the sign convention
//     of MotionShiftSequence; everything else stays with the tile kernel), for which the reference's per-stage clips
//     (SURVEY.md section 8a') reduce to (a) zero-filled x outside the image -- the requests simply skip those cells --,
//     (b) residuals that do not exist: LR row -1 / hl (uniform per wave row) and LR column -1 / wl (lane 0 / 63 of the
//     first / last strip), (c) window taps of the regulariser beyond the right / bottom edge.  (b) and (c) are handled by
//     a few selects in the FIX instances of phase 1; filter2D's dropped blur taps of LR row / column 0 fall on
//     zero-filled x and need nothing.
#pragma once
#include "ztile_dev.hpp"

namespace srmap {
namespace {

template <typename T, int S, int B, int REGK, int R, int NW_>
struct MCfg {
  using Z = ZCfg<T, S, B, REGK, R>;
  static constexpr int NW = NW_;              // waves = HR rows per step (16: one workgroup per CU; 8: several)
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
#ifndef SRMAP_EXP_MTILE
#define SRMAP_EXP_MTILE 0   // measurement: 8-row bands of ONE step (a tile decomposition on the marching kernel's data path)
#endif
  static constexpr int NXR = SRMAP_EXP_MTILE ? XWIN : XWIN + SR;   // ring: window + the next step's SR new rows
  static constexpr int ZRG = PL * CW;
  static constexpr int NZR = (B > 1) ? SR + 2 * HB : 1;
  static constexpr int CCL = 0, CC = CW;          // the left halo columns live in their own small ring (hs)
  static constexpr int CRG = PL * CC;
  static constexpr int NHR = 3 * SR;              // halo ring rows: this step's SR + RU and the next step's SR; SR | NHR
  static constexpr int NCR = REGK ? SR + RU : 1;
  static constexpr int VW = zmax(RU, HB + ZA);    // waves of the virtual step: SR - VW .. SR - 1
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

template <typename T> __device__ __forceinline__ T m_scale() { return Pre<T>::up(T(1)); }

// Pin a set of accumulators at this point of the program: what was computed into them so far is complete here, and no
// memory access moves across (the compiler otherwise gathers the LDS reads of ALL window rows of a pass at its head --
// 80 registers of window data -- and spills around them; __builtin_amdgcn_sched_barrier alone did not stop it).
template <typename T, int N>
__device__ __forceinline__ void pin(T (&a)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(a[i]) : : "memory");
}

// What phase 1 of one wave row has to know beyond the interior case.
struct P1Ctl {
  bool do_z, do_r;     // evaluate the data / regulariser part (ALLON instances: both true at compile time)
  bool count_z;        // the residual row belongs to this band (its cost is counted here)
  bool full;           // the regulariser row belongs to this band (self term, cost); else 2*lambda*w*r only (halo rows)
  int badbits;         // PER LANE: bit v set = the residual of pixel v of this thread's NV does not exist (LR row -1 / hl:
                       // the same bits in every lane; LR column -1 / wl: lane 0 / 63 of the first / last strip)
  bool zero00;         // this row is image row 0 of the first strip: lane 0's first pixel is the absolute pixel (0,0)
  int gr, H;
};

// ---- phase 1 of one wave: regulariser pass 1 for HR row gr and data term (B x, residuals, z, horizontal half of B^T)
// for HR row zrow = gr + ZA, from the SAME window rows (x rows gr .. gr + WIN).
//   xb[i]   first granule (+ lane) of x row gr + i
//   zdst    first granule (+ lane) of the zh row's slot,  cdst: of the 2*lambda*w*r row's slot
//   SCm     2^Q, but 0 in lane 63 of the last strip: window taps that cross into the next cell form their difference
//           with it (and with x0 * SCm), so they come out as the reference's skipped tap -- a zero difference
// FIX: strips at the image border -- residuals without an LR pixel (badbits), window taps beyond the right edge (SCm),
// the absolute pixel (0,0).  FIXR (with FIX): rows at the image border too -- window rows below the image, rows above
// it.  ALLON: both parts, every row inside the band.  (See the head of this file.)
template <typename T, int S, int B, int REGK, int R, int NW, bool FIX, bool FIXR, bool ALLON, typename ArgsT>
__device__ __forceinline__ void m_phase1(const ArgsT& A, const typename Gran<T, 16 / (int)sizeof(T)>::type* const (&xb)[MCfg<T, S, B, REGK, R, NW>::WIN + 1],
                                         typename Gran<T, 16 / (int)sizeof(T)>::type* zdst,
                                         typename Gran<T, 16 / (int)sizeof(T)>::type* cdst,
                                         const PWin<T, S, MCfg<T, S, B, REGK, R, NW>::XC, MCfg<T, S, B, REGK, R, NW>::XCL, MCfg<T, S, B, REGK, R, NW>::P1LO, MCfg<T, S, B, REGK, R, NW>::P1HI>& w0,
                                         const P1Ctl& ctl, int lane,
                                         const T SCm, const T (&ypre)[MCfg<T, S, B, REGK, R, NW>::NV], const T (&wv)[S],
                                         T (&acc)[S], T (&zown)[S], double& cost) {
  using C = MCfg<T, S, B, REGK, R, NW>;
  using GT = typename Gran<T, C::G>::type;
  constexpr int HB = C::HB, NV = C::NV, WIN = C::WIN, G = C::G, PL = C::PL;
  const bool do_z = ALLON || ctl.do_z, do_r = ALLON || ctl.do_r;
  const T SC = m_scale<T>();
  // ---- data term first (its registers -- B x at NV pixels, the observations -- are free again before the regulariser
  // pass starts: the window rows are read twice, the kernel stays clear of the 128-register line) ----
  // residuals of the frames whose LR grid hits each pixel (objective_data_term.cpp:15-75)
  if (do_z) {
    T bx[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) bx[v] = T(0);
    // window rows double-buffered by hand, a scheduling barrier per row: left alone the scheduler issues the reads of ALL
    // rows of a pass first (80 registers of window data) and the allocator spills around them
    // (row 0 arrives preloaded -- w0, requested in front of the step barrier)
    PWin<T, S, C::XC, C::XCL, C::P1LO, (S - 1 + 2 * HB)> wd[2];
#pragma unroll
    for (int i = 0; i < B; ++i) {  // blur row a = i of the residual row (x row zrow - HB + a = gr + i)
      if (i + 1 < B) wd[(i + 1) & 1].load(xb[i + 1]);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
#pragma unroll
        for (int e = 0; e < B; ++e) bx[v] += blur_tap<B>(A, i, e) * (i == 0 ? w0.at(v + e - 2 * HB) : wd[i & 1].at(v + e - 2 * HB));
      }
      pin(bx);
    }
    T z[NV];
    double cz = 0.0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int pcv = v - HB, pc = posmod(pcv, S), dc = floordiv(pcv, S);
      const bool own = pcv >= 0 && pcv < S;
      T rr = bx[v] - ypre[v];
      if (FIX) rr = ((ctl.badbits >> v) & 1) ? T(0) : rr;   // no such LR pixel
      z[v] = rr;
      if (own) cz += (double)rr * (double)rr;
    }
    if (ctl.count_z) cost += (double)(S * S) * cz;  // uniform
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
  __builtin_amdgcn_sched_barrier(0);   // (the scheduler would merge the two passes again)
  // ---- regulariser pass 1 (tv_regularizer.cpp:110-170, btv_regularizer.cpp:19-136) ----
  if (REGK != 0 && do_r) {
    T x0s[S], rv[S], dv[S];
#pragma unroll
    for (int pc = 0; pc < S; ++pc) { rv[pc] = T(0); dv[pc] = T(0); x0s[pc] = T(0); }
    PWin<T, S, C::XC, C::XCL, 0, S - 1 + WIN> wr[2];
#pragma unroll
    for (int p0 = 0; p0 <= S - 1 + WIN; ++p0) wr[0].g[floordiv(p0, G) - wr[0].GLO][posmod(p0, G)] = w0.at(p0);   // (register renaming)
#pragma unroll
    for (int i = 0; i <= WIN; ++i) {
      if (i + 1 <= WIN) wr[(i + 1) & 1].load(xb[i + 1]);   // (rows below the image hold zeros: read, not used)
      if (FIXR && ctl.gr + i >= ctl.H) {
        // window row below the image: its taps are the reference's skipped taps = zero differences; the self term in
        // its (sgn + 1) / 2 form still counts them with 1/2 each (the same additions as the tile kernel's masked path)
        if (REGK == 2 && sizeof(T) == 8 && i < R) {
#pragma unroll
          for (int pc = 0; pc < S; ++pc) {
#pragma unroll
            for (int j = 0; j < R; ++j) dv[pc] += A.powtab[i + j] * T(0.5);
          }
        }
        continue;
      }
      const auto& w = wr[i & 1];
      if (i == 0) {
#pragma unroll
        for (int pc = 0; pc < S; ++pc) x0s[pc] = w.at(pc) * SC;
      }
      // taps that cross into the next cell: in lane 63 of the last strip they lie beyond the image = the reference's
      // skipped taps = zero differences; formed with the masked scale SCm (0 there, 2^Q elsewhere) they come out so
      T mR = T(1);
      if (FIX) mR = Pre<T>::down(SCm);
#pragma unroll
      for (int pc = 0; pc < S; ++pc) {
        T x0m = x0s[pc];
        if (FIX && pc + (REGK == 2 ? R : 1) >= S) x0m = x0s[pc] * mR;
        if (REGK == 2) {
#pragma unroll
          for (int j = 0; j <= R; ++j) {
            if (i == 0 && j == 0) continue;
            // (x[p] - x[q]) * 2^Q, one rounding
            const T d = (FIX && pc + j >= S) ? __builtin_fma(-SCm, w.at(pc + j), x0m) : __builtin_fma(-SC, w.at(pc + j), x0s[pc]);
            rv[pc] += A.powtab[i + j] * absv(d);
            if (i < R && j < R) {
              if (sizeof(T) == 8) dv[pc] += A.powtab[i + j] * step_pre<T>(d);
              else dv[pc] += sgn_pre<T>(d, A.powtab[i + j]);
            }
          }
        } else if (i == 1) {
          const T dyv = __builtin_fma(SC, w.at(pc), -x0s[pc]);
          rv[pc] = absv(dyv) + rv[pc];
          dv[pc] = dv[pc] - sgn_pre<T>(dyv, T(1));
        } else if (i == 0) {
          const T dxv = (FIX && pc + 1 >= S) ? __builtin_fma(SCm, w.at(pc + 1), -x0m) : __builtin_fma(SC, w.at(pc + 1), -x0s[pc]);
          rv[pc] = absv(dxv);
          dv[pc] = -sgn_pre<T>(dxv, T(1));
        }
      }
      pin(rv);
      pin(dv);
    }
    // 2*lambda*w*r, self term, cost
    T cr2v[S];
    const bool full = ALLON || ctl.full;
#pragma unroll
    for (int pc = 0; pc < S; ++pc) {
      const T r = Pre<T>::down(rv[pc]);
      const T c = A.lambda * wv[pc];
      const T cr2 = T(2) * c * r;
      if (REGK == 2 && sizeof(T) == 8) dv[pc] = T(2) * dv[pc] - A.pwsum;
      if (full) {  // uniform
        acc[pc] += cr2 * dv[pc];
        cost += (double)c * (double)r * (double)r;
      }
      cr2v[pc] = cr2;
    }
    if (FIX) {
      if (FIXR && ctl.gr < 0) {  // rows above the image (halo rows of the top band)
#pragma unroll
        for (int pc = 0; pc < S; ++pc) cr2v[pc] = T(0);
      }
      // the absolute pixel (0,0) is skipped as a source (btv_regularizer.cpp:143-146)
      if (REGK == 2 && ctl.zero00) cr2v[0] = (lane == 0) ? T(0) : cr2v[0];
    }
#pragma unroll
    for (int pl = 0; pl < PL; ++pl) {
      GT o;
#pragma unroll
      for (int e = 0; e < G; ++e) o[e] = cr2v[pl * G + e];
      cdst[pl * C::CC + C::CCL] = o;
    }
  }
}

// 2*lambda*w*r of ONE left-halo-column pixel (column COL < 0 relative to the strip), one row per lane.
//   xe[i]  per-lane ELEMENT offset of x row gr + i's slot,  ce: of the 2*lambda*w*r row's slot (xs / cs as T arrays)
template <typename T, int S, int B, int REGK, int R, int NW, int COL, bool BORDER, typename ArgsT>
__device__ __forceinline__ void m_halo_col(const ArgsT& A, const T* __restrict__ xs, T* __restrict__ cs, T wt,
                                           const int (&xe)[MCfg<T, S, B, REGK, R, NW>::WIN + 1], int ce, int gr, int gc) {
  using C = MCfg<T, S, B, REGK, R, NW>;
  constexpr int WIN = C::WIN, G = C::G;
  auto xel = [](int p) constexpr { return ((posmod(floordiv(p, G) * G, S) / G) * C::XC + C::XCL + floordiv(floordiv(p, G) * G, S)) * G + posmod(p, G); };
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
  cs[ce + posmod(COL, G)] = cr2;   // halo ring: one granule per row, the pixel's place inside its granule
}

// ---- phase 2 of one wave (HR row gr): vertical half of B^T, regulariser pass 2 (tv_regularizer.cpp:172-203,
// btv_regularizer.cpp:137-162); acc holds the self term of pass 1 on entry, the gradient on exit ----
//   xb[i] / cb[i]  first granule (+ lane) of x / 2*lambda*w*r row gr - i;  zb[a]: of zh row gr - HB + a;  hb[i]: the halo
//   ring's granule of row gr - i (pixels left of the strip)
template <typename T, int S, int B, int REGK, int R, int NW, typename ArgsT, typename Mid>
__device__ __forceinline__ void m_phase2(const ArgsT& A, const typename Gran<T, 16 / (int)sizeof(T)>::type* const (&xb)[MCfg<T, S, B, REGK, R, NW>::RU + 1],
                                         const typename Gran<T, 16 / (int)sizeof(T)>::type* const (&cb)[MCfg<T, S, B, REGK, R, NW>::RU + 1],
                                         const typename Gran<T, 16 / (int)sizeof(T)>::type* const (&hb)[MCfg<T, S, B, REGK, R, NW>::RU + 1],
                                         const typename Gran<T, 16 / (int)sizeof(T)>::type* const (&zb)[B],
                                         const PWin<T, S, MCfg<T, S, B, REGK, R, NW>::XC, MCfg<T, S, B, REGK, R, NW>::XCL, MCfg<T, S, B, REGK, R, NW>::P2LO, MCfg<T, S, B, REGK, R, NW>::P2HI>& xw0,
                                         int lane, bool want_data, bool want_reg,
                                         const T (&zown)[S], T (&acc)[S], Mid&& mid) {
  using C = MCfg<T, S, B, REGK, R, NW>;
  constexpr int RU = C::RU;
  const T SC = m_scale<T>();
  const bool reg_on = REGK != 0 && RU > 0 && want_reg;
  // every read of the data part and row 0 of the regulariser part go out together (one LDS round trip, not four)
  PWin<T, S, C::CW, 0, 0, S - 1> zw[B];
  PWin<T, S, C::XC, C::XCL, C::P2LO, C::P2HI> xw;
  PWin<T, S, C::CC, C::CCL, C::P2LO, C::P2HI> cw;
  // the granule left of the strip (lane 0) comes from the halo ring: one select on the address
  static_assert(RU == 0 || (C::P2LO >= -C::G && C::P2LO < 0), "the left window reaches one granule into the neighbour cell");
  auto load_c = [&](int i, PWin<T, S, C::CC, C::CCL, C::P2LO, C::P2HI>& w) {
    w.g[0] = *((lane == 0) ? hb[i] : cb[i] + w.gidx(w.GLO));
#pragma unroll
    for (int gi = 1; gi < w.NG; ++gi) w.g[gi] = cb[i][w.gidx(w.GLO + gi)];
  };
  if (B > 1 && want_data) {
#pragma unroll
    for (int a = 0; a < B; ++a) zw[a].load(zb[a]);
  }
  if (reg_on) load_c(0, cw);
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
#pragma unroll
        for (int pc = 0; pc < S; ++pc) zz[pc] += k1_tap<B>(A, a) * zw[a].at(pc);
      }
    }
#pragma unroll
    for (int pc = 0; pc < S; ++pc) acc[pc] += sc * zz[pc];
  }
  pin(acc);
  mid();
  if (reg_on) {
    T x0s[S], sum[S];
#pragma unroll
    for (int pc = 0; pc < S; ++pc) { sum[pc] = T(0); x0s[pc] = xw0.at(pc) * SC; }
#pragma unroll
    for (int i = 0; i <= RU; ++i) {
      if (i > 0) { xw.load(xb[i]); load_c(i, cw); }
#pragma unroll
      for (int pc = 0; pc < S; ++pc) {
        if (REGK == 2) {
          if (i < R) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
              if (i == 0 && j == 0) continue;
              // -sgn(x[q] - x[p]) * alpha^(i+j) * 2 c[q] r[q],  q = p - (i, j)
              const T xq = (i == 0) ? xw0.at(pc - j) : xw.at(pc - j);
              sum[pc] += cw.at(pc - j) * sgn_pre<T>(__builtin_fma(-SC, xq, x0s[pc]), A.powtab[i + j]);
            }
          }
        } else {
          if (i == 0) sum[pc] += cw.at(pc - 1) * sgn_pre<T>(__builtin_fma(-SC, xw0.at(pc - 1), x0s[pc]), T(1));
          else sum[pc] += cw.at(pc) * sgn_pre<T>(__builtin_fma(-SC, xw.at(pc), x0s[pc]), T(1));
        }
      }
      pin(sum);
    }
#pragma unroll
    for (int pc = 0; pc < S; ++pc) acc[pc] += sum[pc];
  }
}

}  // namespace
}  // namespace srmap
