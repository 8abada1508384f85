// kernels_march.hip -- the hot path as MARCHING WAVES: one MAP gradient iteration
// (ObjectiveFunction::ComputeAllTerms, objective_function.cpp:5-20) in ONE launch, no workgroup barriers.
//
// Same formulation as kernels_ztile.hip (owner computes on the HR grid, DESIGN.md section 3.1: data term
// objective_data_term.cpp:15-116 over image_model.cpp:86-101; TV tv_regularizer.cpp:110-227; BTV
// btv_regularizer.cpp:19-170), different work decomposition.  A workgroup IS one wave.  It owns a strip of 64 LR cells
// (64 * S HR columns, lane = cell, a thread owns the S pixels of its cell) and marches down a band of RB HR rows, one
// row per iteration.  The rows of x it needs (t-RU .. t+HD around the output row t) live in a wave-private LDS ring
// in the tile kernel's polyphase layout; 2*lambda*w*r (the regulariser's pass-1 result) in a second ring; the
// horizontally blurred residual rows zh in registers.  Per iteration:
//     requests for the NEXT iteration (x row t+HD+1, IRLS weights, observations) are issued first,
//     regulariser pass 1 of row t          (rows t .. t+WIN of the ring)   -> self term, cost, 2*lambda*w*r -> ring,
//     residual row t+HB                    (rows t .. t+2HB)               -> zh (registers),
//     row t: vertical half of B^T, regulariser pass 2, border correction, g store,
//     the prefetched x row replaces the oldest ring row.
// Nothing is shared between waves, so there is no barrier, no lock-step between co-resident waves, no load phase
// (a wave's memory requests are always one iteration ahead of its arithmetic) and no second launch:
//   * the 2*lambda*w*r values of the RU pixel columns left of the strip (owned by the neighbour strip) are
//     evaluated once per band from global memory, one pixel per lane, and parked in LDS;
//   * the border tasks (kernels_ztile.hip "border blocks": what the frame-summed path cannot express at the image
//     border, with the reference's literal per-frame formulas) are carried by the first waves of the grid while their
//     own x rows are in flight; waves that own border pixels wait for them (device-scope counter) before their first
//     output row and subtract the corrections before they store g;
//   * every wave publishes its cost partial with a write-through store and draws a ticket; the last arriver reduces
//     the partials in index order (deterministic) and publishes the cost (and g.d) -- k_finish_eval's job.
// No MFMA: stencil path.
#include "ztile_dev.hpp"

namespace srmap {

namespace {

constexpr int kMarchMaxRB = 64;   // rows per band (the left-halo-column table in LDS is sized for it)
constexpr int kMarchMinRB = 4;

template <typename T, int S, int B, int REGK, int R>
struct MCfg {
  static constexpr int CW = 64;                // LR cells per strip = lanes
  static constexpr int TW = CW * S;
  static constexpr int HB = (B - 1) / 2;
  static constexpr int WIN = REGK == 2 ? R : (REGK == 1 ? 1 : 0);      // pass 1 reaches WIN pixels right / down
  static constexpr int RU = REGK == 2 ? R - 1 : (REGK == 1 ? 1 : 0);   // pass 2 reaches RU pixels up / left
  static constexpr int HD = zmax(WIN, 2 * HB);                          // x rows below the output row
  static constexpr int NRX = RU + 1 + HD;                               // ring rows of x: t-RU .. t+HD
  static constexpr int XCL = zceil(zmax(RU, 2 * HB), S);                // x halo cells left / right
  static constexpr int XCR = zceil(zmax(WIN, 2 * HB), S);
  static constexpr int XC = CW + XCL + XCR;
  static constexpr int XROW = S * XC;
  static constexpr int NV = S + 2 * HB;        // pixels a thread evaluates B x / z at: own S + HB each side
  static constexpr int CCL = RU > 0 ? zceil(RU, S) : 0;
  static constexpr int CC = CW + CCL;
  static constexpr int CROW = S * CC;
  static constexpr int NRC = REGK ? RU + 1 : 0;                         // ring rows of 2*lambda*w*r: t-RU .. t
  static constexpr int NP = REGK == 2 ? 2 * R + 1 : 1;
  static constexpr int NZ = 2 * HB + 1;                                 // zh rows held in registers: t-HB .. t+HB
  static constexpr int PRE = zmax(RU, 2 * HB);                          // iterations before the band's first output row
  static constexpr int XS_ELEMS = NRX * XROW;
  static constexpr int CS_ELEMS = NRC * CROW;
  static constexpr int CSH_ELEMS = RU > 0 ? (kMarchMaxRB + RU) * RU : 0;
  static constexpr bool REGK2 = REGK == 2;
};

template <typename T, int B, int NP>
struct MArgs : ZArgs<T, B, NP> {
  int RB;               // rows per band, interior strips
  int nbands;           // bands per interior strip
  int RBe;              // rows per band of the first and the last strip (their masked code paths cost more per row)
  int nbe;              // bands per edge strip; gridDim.x = 2 * nbe + (nstrips - 2) * nbands, gridDim.y = channels
  int nstrips;
  int nduty;            // the first nduty waves of the grid carry the border tasks
  int ntasks;           // border tasks per channel (64 pixels of the border frame each)
  int ntasks_total;     // ... of the evaluation
  int n_wave_partials;  // cost partials: one per wave, then one per border task
  int n_partials;
  int duty_at_end;      // no wave consumes border corrections: the border tasks are picked up by waves that have finished
  unsigned long long task_base;  // value of the task counter (ctr64) at launch
  unsigned long long* ctr64;     // monotonic task counter of the end-of-wave pick-up
  int one_round;        // every pixel phase owns exactly one residual (K = S*S frames with distinct phases)
  int finish;           // 1: the last arriver reduces the partials into cost_out
  unsigned* ctr;        // [0] ticket, [1] duty waves done, [2] a border wait timed out
  double* cost_out;     // [0] cost, [1] g.d (WD)
  double* pub;          // solver line search: host-mapped {cost, g.d}, then the arrival tag
  double* tag_slot;
  double tag;
  double* mpart;        // finish: write-through cost granules [n_partials], sentinel = not yet published
  double* mpart_gd;     //         the same for g.d [n_wave_partials]
  unsigned long long* dbg;  // development builds: per-wave time stamps (nullptr otherwise)
};

// ---- index helpers: `col` is a pixel column relative to the first pixel of the thread's cell ----
template <typename C>
__device__ __forceinline__ constexpr int mxi(int col) {
  return posmod(col, C::TW / C::CW) * C::XC + C::XCL + floordiv(col, C::TW / C::CW);
}
template <typename C>
__device__ __forceinline__ constexpr int mci(int col) {
  return posmod(col, C::TW / C::CW) * C::CC + C::CCL + floordiv(col, C::TW / C::CW);
}

// Phase boundary: the values are formed HERE, before any later memory operation, and no later memory operation moves up
// past this point.  Left alone the compiler sinks a phase's arithmetic below the next phase's LDS reads (and hoists those
// reads to the top of the iteration), the iteration's live values exceed the register file and the prefetched inputs of
// the next iteration go to scratch memory behind an s_waitcnt on requests that were issued a moment ago.
template <typename T, int N>
__device__ __forceinline__ void m_pin(T (&a)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(a[i]) : : "memory");
}

// The frame table's by-value part (counts, round 0) is indexed with the row phase of the current row: a run-time index.
// Reading it through `A` makes the argument block an indexed aggregate, and once the kernel is large the compiler keeps
// a private copy of the WHOLE block in scratch memory (every argument then costs a scratch load).  These accessors read
// the kernel-argument segment directly (uniform address -> scalar load); `A` itself is only ever indexed statically.
template <typename ArgsT, typename U>
__device__ __forceinline__ U m_karg(size_t byte_off) {
  typedef const char __attribute__((address_space(4))) * CP;
  typedef const U __attribute__((address_space(4))) * UP;
  CP base = (CP)__builtin_amdgcn_kernarg_segment_ptr();
  return *(UP)(base + byte_off);
}
template <typename ArgsT>
__device__ __forceinline__ int m_cntk(int pr, int i) {
  return m_karg<ArgsT, int>(__builtin_offsetof(ArgsT, cntk) + (size_t)(pr * 8 + i) * sizeof(int));
}
template <typename ArgsT>
__device__ __forceinline__ long long m_off0(int pr, int pc) {
  return m_karg<ArgsT, long long>(__builtin_offsetof(ArgsT, off0) + (size_t)(pr * 4 + pc) * sizeof(long long));
}
template <typename ArgsT>
__device__ __forceinline__ ZEntry m_aux0(int pr, int pc) {
  const size_t o = __builtin_offsetof(ArgsT, aux0) + (size_t)(pr * 4 + pc) * sizeof(ZEntry);
  ZEntry e;
  e.k = m_karg<ArgsT, int>(o); e.io = m_karg<ArgsT, int>(o + 4); e.jo = m_karg<ArgsT, int>(o + 8); e.oyx = m_karg<ArgsT, int>(o + 12);
  return e;
}

// Observations of the t-th residual of each of the NV pixels of the thread's cell in an HR row of phase pr
// (ztile_dev.hpp load_obs_row, with the by-value table read through the accessors above).
template <typename T, int S, typename C, bool EDGE, typename ArgsT>
__device__ __forceinline__ void m_load_obs_row(const ArgsT& A, int pr, int rc, int t, int cell0, int lane,
                                               const T* __restrict__ ybase, T (&yv)[C::NV]) {
  constexpr int HB = C::HB, NV = C::NV;
  const size_t slot = (size_t)(t * S + pr) * S;  // uniform; round-major: round 0 (the prefetch) needs no table size
  const T* yrow = ybase + ((long long)rc * A.wl + cell0);  // uniform: LR cell row rc, first cell of the strip
  if (!EDGE) {
    long long offs[S];
#pragma unroll
    for (int pc = 0; pc < S; ++pc) offs[pc] = (t == 0) ? m_off0<ArgsT>(pr, pc) : ctab(A.off, slot + pc);  // t: uniform
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int pcv = v - HB, pc = posmod(pcv, S), dc = floordiv(pcv, S);
      const T* yp = yrow + (offs[pc] + dc);  // uniform pointer; the lane adds its (non-negative) cell index
      yv[v] = yp[(unsigned)lane];
    }
    return;
  }
  ZEntry ent[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) ent[pc] = (t == 0) ? m_aux0<ArgsT>(pr, pc) : ctab(A.aux, slot + pc);  // t: uniform
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int pcv = v - HB, pc = posmod(pcv, S), dc = floordiv(pcv, S);
    const ZEntry e = ent[pc];
    yv[v] = obs_at<T>(ybase + (size_t)e.k * A.obs_C * ((size_t)A.wl * A.hl), rc + e.io, cell0 + lane + dc + e.jo, A.hl, A.wl);
  }
}

// ---- data term for the S pixels of one cell in HR row zr (wave-uniform) ----
// B x at NV pixels, residuals of the frames whose LR grid hits each pixel, z; returns z (B == 1) or the horizontal
// half of B^T z (B == 3).  xr[a]: ring row zr - HB + a (lane folded in).  `count`: the row belongs to this band.
// EDGE: bands near the image border -- LR validity masks, in-image masks and the dropped blur taps of LR row 0 /
// column 0 (kernels_ztile.hip z_row).  The staged x is pre-scaled by 2^Q: residual = (B x') * 2^-Q - y.
// ONE: every (row phase, column phase) owns exactly one residual (K = S*S frames with distinct phases): one round, no loop.
// DM: 0 interior, 1 only COLUMNS can leave the LR image (first / last strips away from the image top / bottom; ONE only),
// 2 the general EDGE path.
template <typename T, int S, int B, typename C, int DM, bool ONE, typename ArgsT>
__device__ __forceinline__ void mz_row(const ArgsT& A, const T* const (&xr)[C::NZ], int zr, int cell0, int lane,
                                       const T* __restrict__ ybase, const T (&ypre)[C::NV], bool count,
                                       const T (&mk)[S], T (&zout)[S], double& cost) {
  constexpr int HB = C::HB, NV = C::NV;
  constexpr bool EDGE = DM == 2;
  static_assert(DM != 1 || ONE, "the column-edge path serves one residual per pixel phase");
  int rc, pr;
  row_phase<S>(zr, rc, pr);
  T bx[NV], btop[NV], bleft[NV], bcorner[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) { bx[v] = T(0); btop[v] = T(0); bleft[v] = T(0); bcorner[v] = T(0); }
  {
    // one window row of look-ahead, pinned: left alone the scheduler requests every row of every phase up front and
    // the prefetched inputs of the next iteration end up in scratch memory
    T xv[B][NV + B - 1];
#pragma unroll
    for (int j = 0; j < NV + B - 1; ++j) xv[0][j] = xr[0][mxi<C>(j - 2 * HB)];
#pragma unroll
    for (int a = 0; a < B; ++a) {
      if (a + 1 < B) {
#pragma unroll
        for (int j = 0; j < NV + B - 1; ++j) xv[a + 1][j] = xr[a + 1][mxi<C>(j - 2 * HB)];
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) {
#pragma unroll
        for (int e = 0; e < B; ++e) bx[v] += blur_tap<B>(A, a, e) * xv[a][v + e];
      }
    }
  }
  // filter2D's zero padding acts on the WARPED image: a residual of LR row 0 loses blur tap row 0, of LR column 0 tap
  // column 0.  LR row 0 can only be met by the rows next to the image top, LR column 0 by the leftmost strips (uniform).
  // A frame with offset o_k <= 0 needs nothing: the dropped tap reads x outside the image, staged as 0.  Frames with a
  // positive offset component exist exactly when the border frame has in-image rectangles (ring.rg[0] / rg[1] > 0).
  const bool need_top = EDGE && B > 1 && A.ring.rg[0] > 0 && rc <= 1 + A.E / S;
  const bool need_left = EDGE && B > 1 && A.ring.rg[1] > 0 && cell0 <= 1 + A.E / S;
  if (need_top || need_left) {
#pragma unroll
    for (int a = 0; a < B; ++a) {
      T xv[NV + B - 1];
#pragma unroll
      for (int j = 0; j < NV + B - 1; ++j) xv[j] = xr[a][mxi<C>(j - 2 * HB)];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        bleft[v] += blur_tap<B>(A, a, 0) * xv[v];  // tap column 0
        if (a == 0) {
#pragma unroll
          for (int e = 0; e < B; ++e) btop[v] += blur_tap<B>(A, 0, e) * xv[v + e];  // tap row 0
          bcorner[v] = blur_tap<B>(A, 0, 0) * xv[v];
        }
      }
    }
  }
  const T unscale = Pre<T>::down(T(1));
  if (DM == 1) {
    // rows are interior: every (frame, LR row) of the table exists; an LR column outside the image (per lane) is a
    // zero residual.  No dropped blur taps unless a frame has a positive column offset (need_left: general path).
    T z[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int pcv = v - HB, pc = posmod(pcv, S), dc = floordiv(pcv, S);
      const int j = cell0 + lane + dc + m_aux0<ArgsT>(pr, pc).jo;
      T rr = bx[v] * unscale - ypre[v];
      rr = ((unsigned)j < (unsigned)A.wl) ? rr : T(0);
      z[v] = rr;
      if (pcv >= 0 && pcv < S && count) cost += (double)(rr * mk[pcv >= 0 && pcv < S ? pcv : 0]) * (double)rr;
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
        zout[pc] = zh;
      }
    }
    return;
  }
  int cn[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) cn[pc] = ONE ? 1 : m_cntk<ArgsT>(pr, pc);
  const int mmax = ONE ? 1 : m_cntk<ArgsT>(pr, S);
  const int mfull = EDGE ? 0 : (ONE ? 1 : m_cntk<ArgsT>(pr, S + 1));  // rounds in which every column phase owns a residual
  T z[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) z[v] = T(0);
#pragma unroll 1
  for (int t = 0; t < mmax; ++t) {
    T yv[NV];
    if (t == 0) {
#pragma unroll
      for (int v = 0; v < NV; ++v) yv[v] = ypre[v];
    } else {
      m_load_obs_row<T, S, C, EDGE>(A, pr, rc, t, cell0, lane, ybase, yv);
    }
    if (!EDGE && t < mfull) {  // uniform; the common case (K a multiple of S*S distinct phases): no per-pixel selects
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
    if (!EDGE) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int pcv = v - HB, pc = posmod(pcv, S);
        if (t < cn[pc]) {  // uniform
          const T rr = bx[v] * unscale - yv[v];
          z[v] += rr;
          if (pcv >= 0 && pcv < S && count) cost += (double)rr * (double)rr;
        }
      }
      continue;
    }
    // EDGE, without branches: (frame, LR row, LR column) of every slot; a slot beyond the phase's count, an LR row
    // outside the image (both uniform) or an LR column outside it (per lane) contributes a zero residual.
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int pcv = v - HB, pc = posmod(pcv, S), dc = floordiv(pcv, S);
      const bool own = pcv >= 0 && pcv < S;
      const ZEntry e = (t == 0) ? m_aux0<ArgsT>(pr, pc) : ctab(A.aux, slot + pc);
      const int i = rc + e.io, j = cell0 + lane + dc + e.jo;
      const bool row_ok = (ONE || t < cn[pc]) && (unsigned)i < (unsigned)A.hl;  // uniform
      T bxv = bx[v];
      if (B > 1 && (need_top || need_left)) {  // uniform: rows next to the image top / the leftmost strips only
        const bool j0 = j == 0;
        if (i == 0) bxv = bxv - btop[v] - (j0 ? bleft[v] - bcorner[v] : T(0));
        else bxv = bxv - (j0 ? bleft[v] : T(0));
      }
      T rr = bxv * unscale - yv[v];
      rr = (row_ok && (unsigned)j < (unsigned)A.wl) ? rr : T(0);
      z[v] += rr;
      if (own) {
        const bool cnt_ok = count && S * i >= A.cr0 && S * i < A.cr1;  // uniform
        const double rd = (double)(rr * (cnt_ok ? mk[pcv >= 0 && pcv < S ? pcv : 0] : T(0)));
        cost += rd * (double)rr;
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
      zout[pc] = zh;
    }
  }
}

// ---- regulariser pass 1 for the S pixels of one cell in HR row gr (tv_regularizer.cpp:110-170,
// btv_regularizer.cpp:19-136) ----
// xr[i]: ring row gr + i.  FULL: values, self term into acc, cost, 2*lambda*w*r into csrow.  !FULL: 2*lambda*w*r only
// (the RU rows above the band).  BORDER: the window can leave the image at the right / bottom edge (skipped taps),
// pixels right of the image store 0.  zero00: this wave holds the absolute pixel (0,0), whose 2*lambda*w*r is never
// propagated (btv_regularizer.cpp:143-146).
template <typename T, int S, int REGK, int R, typename C, bool BORDER, bool FULL>
__device__ __forceinline__ void mreg_row(T (&acc)[S], double& cost, const T* const (&xr)[C::WIN + 1], T* __restrict__ csrow,
                                         const T (&c2v)[S], int lane, int gr, int gc0, int W, int H,
                                         const T (&pw)[C::NP], T pwsum, bool cost_row, bool zero00,
                                         const T (&cmk)[C::WIN > 0 ? C::WIN : 1]) {
  // c2v = 2 * (lambda * w): the caller forms it when the weights arrive, so that their registers can take the next
  // row's request at once.  2 c r = c2v r and c r^2 = (c2v r) r / 2 exactly (powers of two).
  // BORDER masks: a window ROW below the image is skipped as a whole (uniform); a window COLUMN right of the image
  // can only be one of the WIN columns behind the thread's own cell: cmk[c] = 1 / 0 for relative column S + c (a
  // multiply on the difference: skipped tap == zero difference).
  constexpr int WIN = C::WIN;
  constexpr int NC = S + WIN;
  T x0v[S], rv[S], dv[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) { rv[pc] = T(0); dv[pc] = T(0); }
  // the window is walked row by row (i outer, j inner per pixel: the reference's summation order), the next row's LDS
  // reads one row ahead of the arithmetic and pinned there (see mz_row)
  T rows[WIN + 1][NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) rows[0][j] = xr[0][mxi<C>(j)];
#pragma unroll
  for (int i = 0; i <= WIN; ++i) {
    if (i < WIN) {
#pragma unroll
      for (int j = 0; j < NC; ++j) rows[i + 1][j] = xr[i + 1][mxi<C>(j)];
    }
    if (BORDER && i > 0 && gr + i >= H) {  // uniform
      if (REGK == 2 && FULL && i < R && sizeof(T) == 8) {
        // zero differences: (sgn + 1) / 2 = 1 / 2 for each of the row's taps inside the gradient's window
        T half = T(0);
#pragma unroll
        for (int j = 0; j < R; ++j) half += T(0.5) * pw[i + j];
#pragma unroll
        for (int pc = 0; pc < S; ++pc) dv[pc] += half;
      }
      continue;
    }
    const T (&row)[NC] = rows[i];
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
          if (BORDER && pc + j >= S) d *= cmk[pc + j - S];
          rv[pc] += pw[i + j] * absv(d);
          if (FULL && i < R && j < R) {  // exclusive window in the gradient
            if (sizeof(T) == 8) dv[pc] += pw[i + j] * step_pre<T>(d);  // (sgn + 1) / 2: add with clamp + FMA
            else dv[pc] += sgn_pre<T>(d, pw[i + j]);
          }
        }
      } else if (i == 1) {
        const T dyv = row[pc] - x0v[pc];
        rv[pc] = absv(dyv) + rv[pc];
        if (FULL) dv[pc] = dv[pc] - sgn_pre<T>(dyv, T(1));
      } else {
        T dxv = row[pc + 1] - x0v[pc];
        if (BORDER && pc + 1 >= S) dxv *= cmk[pc + 1 - S];
        rv[pc] = absv(dxv);
        if (FULL) dv[pc] = -sgn_pre<T>(dxv, T(1));
      }
    }
  }
#pragma unroll
  for (int pc = 0; pc < S; ++pc) {
    const T r = Pre<T>::down(rv[pc]);  // the staged x is pre-scaled: r = r' * 2^-Q exactly
    T cr2 = c2v[pc] * r;
    const bool in_img = !BORDER || gc0 + pc < W;
    if (FULL) {
      if (REGK == 2 && sizeof(T) == 8) dv[pc] = T(2) * dv[pc] - pwsum;  // sum pw * sgn = 2 * sum pw * (sgn + 1) / 2 - sum pw
      acc[pc] += cr2 * dv[pc];
      const double cd = (in_img && cost_row) ? (double)(T(0.5) * cr2) * (double)r : 0.0;
      cost += cd;
    }
    if (BORDER && !in_img) cr2 = T(0);
    if (REGK == 2 && pc == 0 && zero00) cr2 = (lane == 0) ? T(0) : cr2;
    csrow[mci<C>(pc)] = cr2;
  }
}

// ---- regulariser pass 2: contributions of the up / left neighbours (tv_regularizer.cpp:172-203,
// btv_regularizer.cpp:137-162).  xu[i] / cu[i]: ring rows gr - i of x / of 2*lambda*w*r.
template <typename T, int S, int REGK, int R, typename C>
__device__ __forceinline__ void mreg_pass2(T (&acc)[S], const T* const (&xu)[C::RU + 1], const T* const (&cu)[C::RU + 1],
                                           const T (&pw)[C::NP]) {
  constexpr int RU = C::RU;
  if (RU == 0) return;
  constexpr int NC = S + RU;
  T x0v[S], sum[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) sum[pc] = T(0);
#pragma unroll
  for (int i = 0; i <= RU; ++i) {  // neighbour row r - i (one row at a time: the registers of a second row in flight
                                    // are what pushes the iteration over the register file)
    T xw[NC], cw[NC];               // columns -RU .. S-1
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      xw[j] = xu[i][mxi<C>(j - RU)];
      cw[j] = cu[i][mci<C>(j - RU)];
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
            sum[pc] += cw[pc + RU - j] * sgn_pre<T>(x0v[pc] - xw[pc + RU - j], pw[i + j]);
          }
        }
      } else {
        if (i == 0) sum[pc] += cw[pc + RU - 1] * sgn_pre<T>(x0v[pc] - xw[pc + RU - 1], T(1));
        else sum[pc] += cw[pc + RU] * sgn_pre<T>(x0v[pc] - xw[pc + RU], T(1));
      }
    }
  }
#pragma unroll
  for (int pc = 0; pc < S; ++pc) acc[pc] += sum[pc];
}

// 2*lambda*w*r of the RU pixel columns left of the strip, rows R0-RU .. tend-1: one pixel per lane, straight from
// global memory (un-scaled x: the sums scale exactly), parked in csh[(row - (R0 - RU)) * RU + hc], hc = 0: column -1.
// Two halves: the requests go out BEFORE the band's x rows (they come back first), the arithmetic runs while those
// rows are still in flight.
template <typename T, typename C>
struct MHalo {
  static constexpr int NVW = C::REGK2 ? (C::WIN + 1) * (C::WIN + 1) : 3;
  T v[NVW];     // window values; [0] = the pixel itself
  T wt;
  unsigned inb;  // bit k: window value k is inside the image
  bool on, live;
};

template <typename T, int S, int REGK, int R, typename C, typename ArgsT>
__device__ __forceinline__ void march_halo_issue(const ArgsT& A, const T* __restrict__ xplane, const T* __restrict__ wplane,
                                                 int R0, int tend, int C0, int lane, int base, MHalo<T, C>& h) {
  constexpr int RU = C::RU, WIN = C::WIN;
  const int nh = (tend - R0 + RU) * RU;
  const int W = A.W, H = A.H;
  const int idx = base + lane;
  const int hr = idx / RU, hc = idx - hr * RU;
  const int gr = R0 - RU + hr, gc = C0 - 1 - hc;
  h.live = idx < nh;
  h.on = idx < nh && gr >= 0 && gr < H && gc >= 0 && gc < W && !(REGK == 2 && gr == 0 && gc == 0);
  const size_t o0 = h.on ? (size_t)gr * W + gc : (size_t)0;
  h.wt = (wplane != nullptr) ? wplane[o0] : T(1);
  h.inb = 0;
  if (REGK == 2) {
#pragma unroll
    for (int i = 0; i <= WIN; ++i)
#pragma unroll
      for (int j = 0; j <= WIN; ++j) {
        const bool ok = h.on && gr + i < H && gc + j < W;
        h.inb |= ok ? (1u << (i * (WIN + 1) + j)) : 0u;
        h.v[i * (WIN + 1) + j] = xplane[ok ? (size_t)(gr + i) * W + (gc + j) : (size_t)0];
      }
  } else {
    const bool oky = h.on && gr + 1 < H, okx = h.on && gc + 1 < W;
    h.inb = (oky ? 2u : 0u) | (okx ? 4u : 0u);
    h.v[0] = xplane[o0];
    h.v[1] = xplane[oky ? (size_t)(gr + 1) * W + gc : (size_t)0];
    h.v[2] = xplane[okx ? (size_t)gr * W + gc + 1 : (size_t)0];
  }
}
template <typename T, int S, int REGK, int R, typename C, typename ArgsT>
__device__ __forceinline__ void march_halo_finish(const ArgsT& A, T* __restrict__ csh, int lane, int base, const MHalo<T, C>& h) {
  constexpr int WIN = C::WIN;
  const T x0 = h.v[0];
  T cr2;
  if (REGK == 2) {
    T r = T(0);
#pragma unroll
    for (int i = 0; i <= WIN; ++i)
#pragma unroll
      for (int j = 0; j <= WIN; ++j) {
        if (i == 0 && j == 0) continue;
        const int k = i * (WIN + 1) + j;
        const T d = ((h.inb >> k) & 1u) ? x0 - h.v[k] : T(0);
        r += A.powtab[i + j] * absv(d);
      }
    cr2 = T(2) * (A.lambda * h.wt) * r;
  } else {
    const T yv = (h.inb & 2u) ? absv(h.v[1] - x0) : T(0);
    const T xv = (h.inb & 4u) ? absv(h.v[2] - x0) : T(0);
    cr2 = T(2) * (A.lambda * h.wt) * (yv + xv);
  }
  if (h.live) csh[base + lane] = h.on ? cr2 : T(0);
}

// Border tasks (kernels_ztile.hip border_block, 64 pixels of the border frame per task).  Corrections and cost partials
// leave with write-through stores.  Two schedules:
//   * some wave consumes corrections (frames with positive offsets): the tasks run as their own launch, k_march_border,
//     before k_eval_march (the kernel boundary orders corrections and their readers);
//   * nobody does (the tasks only produce cost partials): waves that have finished their band pick the tasks up from a
//     counter (k_eval_march, "duty_at_end") -- the earliest finishers absorb them, no wave starts late.
template <typename T, int S>
__device__ __forceinline__ void march_border_tables(const BorderArgs<T>& Bd, int lane, void* smem) {
  int2* s_hdr = reinterpret_cast<int2*>(smem);
  ZEntry* s_ent = reinterpret_cast<ZEntry*>(s_hdr + 16);
  if (lane < S * S) s_hdr[lane] = Bd.hdr[lane];
  for (int i = lane; i < Bd.n_ent; i += 64) s_ent[i] = Bd.ent[i];
  __syncthreads();
}

template <typename T, int S, int B, typename ArgsT>
__device__ __forceinline__ void march_border_task(const ArgsT& A, const BorderArgs<T>& Bd, int task, int lane, void* smem) {
  const int obs_C = Bd.obs_C;
  const int2* s_hdr = reinterpret_cast<const int2*>(smem);
  const ZEntry* s_ent = reinterpret_cast<const ZEntry*>(s_hdr + 16);
  const size_t N = (size_t)A.W * A.H, nl = (size_t)A.wl * A.hl;
  const int ch = task / A.ntasks, bidx = task - ch * A.ntasks;
  const int t = bidx * 64 + lane;
  double cost = 0.0;
  if (t < Bd.n_ring) {
    int qr, qc;
    ring_pixel(t, A.W, A.H, A.ring, qr, qc);
    const T* xplane = A.x + (size_t)ch * N;
    const T* ybase = A.y + (size_t)ch * nl;
    const bool inside = qr >= 0 && qr < A.H && qc >= 0 && qc < A.W;
    T corr = T(0);
    if (!inside) {
      const int rc = dfdiv(qr, S), cc = dfdiv(qc, S);
      const int2 h = s_hdr[(qr - rc * S) * S + (qc - cc * S)];
      for (int n = 0; n < h.x; ++n) {
        const ZEntry e = s_ent[h.y + n];
        const int i = rc + e.io, j = cc + e.jo;
        if (i < 0 || i >= A.hl || j < 0 || j >= A.wl) continue;
        if (S * i < A.cr0 || S * i >= A.cr1) continue;
        const int oy = e.oyx >> 16, ox = (int)(short)(e.oyx & 0xffff);
        const double r = (double)border_residual<T, S, B>(A, A.W, A.H, A.wl, xplane, ybase + (size_t)e.k * obs_C * nl, ox, oy, i, j);
        cost += r * r;
      }
    } else if (A.g != nullptr) {
      constexpr int hb = (B - 1) / 2;
#pragma unroll
      for (int a = 0; a < B; ++a) {
#pragma unroll
        for (int b2 = 0; b2 < B; ++b2) {
          const int pr = qr + a - hb, pc = qc + b2 - hb;
          const int rc = dfdiv(pr, S), cc = dfdiv(pc, S);
          const int2 h = s_hdr[(pr - rc * S) * S + (pc - cc * S)];
          for (int n = 0; n < h.x; ++n) {
            const ZEntry e = s_ent[h.y + n];
            const int oy = e.oyx >> 16, ox = (int)(short)(e.oyx & 0xffff);
            const int ur = qr - oy, uc = qc - ox;
            if (ur >= 0 && ur < A.H && uc >= 0 && uc < A.W) continue;  // frame reaches q: already correct
            const int i = rc + e.io, j = cc + e.jo;
            if (i < 0 || i >= A.hl || j < 0 || j >= A.wl) continue;
            // B^T = correlation with kernel.t() (blur_module.cpp:30-36)
            corr += blur_tap<B>(A, b2, a) *
                    border_residual<T, S, B>(A, A.W, A.H, A.wl, xplane, ybase + (size_t)e.k * obs_C * nl, ox, oy, i, j);
          }
        }
      }
      corr *= (T)(2 * S * S);
    }
    if (A.g != nullptr && inside) st_agent(&Bd.corr[(size_t)ch * Bd.n_ring + t], corr);
  }
  const double cw = wave_sum_d(cost);
  if (lane == 0) st_agent(A.finish ? &A.mpart[(size_t)A.n_wave_partials + task] : &A.partials[(size_t)A.n_wave_partials + task], (double)(S * S) * cw);
}

constexpr int kMarchBorderLds = (int)(16 * sizeof(int2) + kBorderTabEntries * sizeof(ZEntry) + 64);

// What a wave knows about its band (all wave-uniform except lane / gc0).
template <typename T>
struct MBand {
  int lane, ch, R0, tend, CJ0, C0, gc0, t0;
  bool want_data, want_reg, outg, has_ring;
  const T* xplane;
  const T* ybase;
  const T* wplane;
  const T* corr;   // border corrections of this channel (has_ring)
  size_t N;
};

// ---- requests ----
// x row grr -> registers: lane l loads cell CJ0 - XCL + l (S pixels); the EXTRA cells behind those 64 are loaded one
// PIXEL per lane by the first EXTRA * S lanes (contiguous in memory).
// SLOW: rows / cells outside the image read address 0 and are staged as 0 (the warp's zero fill); the scale 2^Q and
// that mask are ONE multiply when the row goes to LDS (a select on the loaded value makes the compiler wait for the
// load where it stands).
template <typename T, int S, typename C, int DM, typename ArgsT>
__device__ __forceinline__ void m_issue_x(const ArgsT& A, const MBand<T>& b, int grr, T (&va)[S], T& vb, T& ma, T& mb) {
  constexpr bool SLOW = DM != 0;
  constexpr int EXTRA = C::XC - C::CW;
  const int gca = b.CJ0 - C::XCL + b.lane;                     // cell of the S-pixel request
  const int hpx = (b.CJ0 - C::XCL + C::CW) * S + b.lane;       // HR column of the halo-pixel request
  const bool hl = b.lane < EXTRA * S;
  if (SLOW) {
    const bool row_in = DM == 1 || (unsigned)grr < (unsigned)A.H;  // uniform (DM 1: the rows are interior)
    const bool ina = row_in && (unsigned)gca < (unsigned)A.wl;
    const bool inb = row_in && hl && hpx < A.W;
    const T* sa = b.xplane + (ina ? (size_t)grr * A.W + (size_t)gca * S : (size_t)0);
    const T* sb = b.xplane + (inb ? (size_t)grr * A.W + (size_t)hpx : (size_t)0);
#pragma unroll
    for (int pc = 0; pc < S; ++pc) va[pc] = sa[pc];
    vb = sb[0];
    ma = ina ? Pre<T>::up(T(1)) : T(0);
    mb = inb ? Pre<T>::up(T(1)) : T(0);
  } else {
    const T* rowp = b.xplane + (size_t)grr * A.W;  // uniform
    const T* sa = rowp + (unsigned)(gca * S);
    const T* sb = rowp + (unsigned)(hl ? hpx : gca * S);  // every lane requests a valid pixel
#pragma unroll
    for (int pc = 0; pc < S; ++pc) va[pc] = sa[pc];
    vb = sb[0];
    ma = Pre<T>::up(T(1));
    mb = ma;
  }
}
template <typename T, int S, typename C>
__device__ __forceinline__ void m_put_x(T* __restrict__ row, int lane, const T (&va)[S], T vb, T ma, T mb) {
  constexpr int EXTRA = C::XC - C::CW;
#pragma unroll
  for (int pc = 0; pc < S; ++pc) row[pc * C::XC + lane] = va[pc] * ma;
  if (lane < EXTRA * S) row[(lane % S) * C::XC + C::CW + lane / S] = vb * mb;
}
template <typename T, int S, int DM, bool SIMPLE, typename ArgsT>
__device__ __forceinline__ void m_issue_w(const ArgsT& A, const MBand<T>& b, int gr, T (&wv)[S]) {
  constexpr bool SLOW = DM != 0;
#pragma unroll
  for (int pc = 0; pc < S; ++pc) wv[pc] = T(1);
  if (!SIMPLE && b.wplane == nullptr) return;  // uniform (SIMPLE: the weights exist)
  // SLOW: rows / lanes outside the image request element 0 (their values are never used): no branch around a load
  const bool ok = !SLOW || ((DM == 1 || (unsigned)gr < (unsigned)A.H) && b.gc0 < A.W);
  const T* wp = b.wplane + (ok ? (size_t)gr * A.W + b.gc0 : (size_t)0);
#pragma unroll
  for (int pc = 0; pc < S; ++pc) wv[pc] = wp[pc];
}
template <typename T, int S, typename C, int DM, typename ArgsT>
__device__ __forceinline__ void m_issue_y(const ArgsT& A, const MBand<T>& b, int zr, T (&yv)[C::NV]) {
  int rc, pr;
  row_phase<S>(zr, rc, pr);
  if (DM == 1) {
    // round 0 only (ONE); (frame, LR row) exist, the LR column is clamped into the image per lane
    constexpr int HB = C::HB;
    const size_t nl = (size_t)A.wl * A.hl;
#pragma unroll
    for (int v = 0; v < C::NV; ++v) {
      const int pcv = v - HB, pc = posmod(pcv, S), dc = floordiv(pcv, S);
      const ZEntry e = m_aux0<ArgsT>(pr, pc);
      const T* rowp = b.ybase + (size_t)e.k * A.obs_C * nl + (size_t)(rc + e.io) * A.wl;  // uniform
      int j = b.CJ0 + b.lane + dc + e.jo;
      j = j < 0 ? 0 : (j >= A.wl ? A.wl - 1 : j);
      yv[v] = rowp[(unsigned)j];
    }
    return;
  }
  m_load_obs_row<T, S, C, DM == 2>(A, pr, rc, 0, b.CJ0, b.lane, b.ybase, yv);
}

// WD: search direction of row gr (zero where the row's terms are not counted / outside the image)
template <typename T, int S, int DM, typename ArgsT>
__device__ __forceinline__ void m_issue_d(const ArgsT& A, const MBand<T>& b, int gr, T (&dv)[S]) {
  constexpr bool SLOW = DM != 0;
  const bool ok = DM == 2 ? ((unsigned)gr < (unsigned)A.H && b.gc0 < A.W && gr >= A.cr0 && gr < A.cr1) : (DM == 1 ? b.gc0 < A.W : true);
  const T* dp = A.dvec + (size_t)b.ch * b.N + (ok ? (size_t)gr * A.W + b.gc0 : (size_t)0);
  const T dm = ok ? T(1) : T(0);
#pragma unroll
  for (int pc = 0; pc < S; ++pc) dv[pc] = SLOW ? dp[pc] * dm : dp[pc];
}

// State a wave carries from row to row.
template <typename T, int S, typename C>
struct MState {
  T w[S];                         // IRLS weights of the next pass-1 row (single buffer: re-requested as soon as consumed)
  T ya[C::NV], yb[C::NV];         // observations of residual rows t+HB and t+1+HB, ping-pong: the buffer a row has consumed takes
                                  // the request for the row after next (a whole iteration ahead of its wait, with the x row); no
                                  // copies -- a rotation of loop-carried values makes the compiler wait for the young request at
                                  // the loop's back edge
  T nva[S], nvb, nma, nmb;        // x row t+HD+1 on its way to the ring
  T p1[S], p2[S];                 // vertical half of B^T in flight: k0 zh[t-1] + k1 zh[t], and k0 zh[t]
  T cmk[C::WIN > 0 ? C::WIN : 1]; // RBD: window columns S .. S+WIN-1 behind the thread's cell, inside the image?
  T mk[S];                        // in-image mask of the thread's pixels
  int ph, phc;                    // ring phases: window row i of x lives in slot (ph + i) mod NRX, of 2*lambda*w*r in (phc + i) mod NRC
  double cost_data, cost_reg, gd;
#ifdef SRMAP_DEV_INSTANCES
  unsigned long long t_wait, t_pass1, t_z, t_p2;  // development: cycles spent waiting at B / in the phases of A
#endif
};

// One row.  The counter of outstanding memory operations drains in order for loads, but not between loads and
// stores: a wait for a load also waits for every store issued before it.  So an iteration has ONE point where it
// waits, for everything, placed where everything is old:
//   A  arithmetic of row t: 2*lambda*w from the weights (their registers take the request for row t+1 at once),
//      residual row t+HB, pass 1, vertical B^T, pass 2;
//      requests consumed at B (search direction, border corrections) go out first
//   B  everything requested so far has landed: x row t+HD+1 -> ring, corrections, g.d
//   C  store g row t
//   D  request x row t+HD+2 and the observations of residual row t+2+HB
// SLOW: masks of the image border (data term EDGE path, lane masks of partial strips, rows / cells outside the image,
// border corrections); RBD: the regulariser's windows leave the image at the right / bottom edge; SIMPLE: data term +
// regulariser + gradient requested, IRLS weights present, one residual per pixel phase (no uniform branches, no loops);
// OUT: row t is an output row of the band (the PRE rows before it only feed the rings).
template <typename T, int S, int B, int REGK, int R, bool WD, int DM, bool RBD, bool SIMPLE, bool OUT, int PP, typename ArgsT>
__device__ __forceinline__ void m_step(const ArgsT& A, const MBand<T>& b, T* __restrict__ xs, T* __restrict__ cs,
                                       const T* __restrict__ csh, MState<T, S, MCfg<T, S, B, REGK, R>>& st, int t) {
  using C = MCfg<T, S, B, REGK, R>;
  constexpr int HB = C::HB, NV = C::NV, RU = C::RU, WIN = C::WIN, HD = C::HD, NRX = C::NRX, NRC = C::NRC, NZ = C::NZ;
  constexpr int PRE = C::PRE;
  constexpr bool SLOW = DM != 0, FULLEDGE = DM == 2;
  const int lane = b.lane;
  const bool want_data = SIMPLE || b.want_data, want_reg = SIMPLE ? (REGK != 0) : b.want_reg, outg = SIMPLE || b.outg;
  const T* xw[NRX];  // rows t-RU .. t+HD, lane folded in
#pragma unroll
  for (int i = 0; i < NRX; ++i) {
    int sl = st.ph + i;
    sl = sl >= NRX ? sl - NRX : sl;
    xw[i] = xs + sl * C::XROW + lane;
  }
  T* cw[NRC > 0 ? NRC : 1];  // rows t-RU .. t
#pragma unroll
  for (int i = 0; i < NRC; ++i) {
    int sl = st.phc + i;
    sl = sl >= NRC ? sl - NRC : sl;
    cw[i] = cs + sl * C::CROW + lane;
  }

#ifdef SRMAP_DEV_INSTANCES
  const unsigned long long ck0 = __builtin_readcyclecounter();
#endif
  // ---------------- A ----------------
  T dreg[S], cq[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) { dreg[pc] = T(0); cq[pc] = T(0); }
  if (WD && OUT) m_issue_d<T, S, DM>(A, b, t, dreg);
  if (FULLEDGE && OUT && b.has_ring) {  // uniform: bands with gradient corrections (top rows / left columns of the image)
#pragma unroll
    for (int pc = 0; pc < S; ++pc) {
      const int ri = (b.gc0 + pc < A.W) ? ring_index(t, b.gc0 + pc, A.W, A.H, A.ring) : -1;
      const T cv = b.corr[ri >= 0 ? ri : 0];  // written by k_march_border, launched before this kernel
      cq[pc] = ri >= 0 ? cv : T(0);
    }
  }
  const bool reg_row = want_reg && (RU == PRE || t >= b.R0 - RU);
  T c2[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) c2[pc] = T(2) * (A.lambda * st.w[pc]);
  if (reg_row) m_issue_w<T, S, DM, SIMPLE>(A, b, t + 1, st.w);  // weights of the next row into the same registers

  T acc[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) acc[pc] = T(0);

  // data term: residual row t + HB, then the vertical half of B^T for row t
  T zz[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) zz[pc] = T(0);
  if (want_data && (2 * HB == PRE || t >= b.R0 - 2 * HB)) {
    const T* xr[NZ];
#pragma unroll
    for (int a = 0; a < NZ; ++a) xr[a] = xw[RU + a];  // rows t .. t + 2 HB
    const int zr = t + HB;
    const bool count = zr >= b.R0 && zr < b.tend;
    T znew[S];
    mz_row<T, S, B, C, DM, SIMPLE>(A, xr, zr, b.CJ0, lane, b.ybase, PP ? st.yb : st.ya, count, st.mk, znew, st.cost_data);
#pragma unroll
    for (int pc = 0; pc < S; ++pc) {
      if (B == 1) {
        zz[pc] = znew[pc];
      } else {  // zz = k0 zh[t-1] + k1 zh[t] + k0 zh[t+1], summed in this order
        zz[pc] = st.p1[pc] + k1_tap<B>(A, 2) * znew[pc];
        st.p1[pc] = st.p2[pc] + k1_tap<B>(A, 1) * znew[pc];
        st.p2[pc] = k1_tap<B>(A, 0) * znew[pc];
      }
    }
    if (OUT && outg) {
      const T sc = (T)(2 * S * S);  // g += 2 * (s*s block sum) (objective_data_term.cpp:55-71)
#pragma unroll
      for (int pc = 0; pc < S; ++pc) acc[pc] = sc * zz[pc];
    }
    m_pin(acc);
    if (B > 1) { m_pin(st.p1); m_pin(st.p2); }
  }

#ifdef SRMAP_DEV_INSTANCES
  const unsigned long long ck1 = __builtin_readcyclecounter();
#endif
  // regulariser pass 1, row t
  if (reg_row) {
    T* csrow = cw[NRC > 0 ? NRC - 1 : 0];
    if (FULLEDGE && t < 0) {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) csrow[mci<C>(pc)] = T(0);
    } else {
      const T* xr[WIN + 1];
#pragma unroll
      for (int i = 0; i <= WIN; ++i) xr[i] = xw[RU + i];
      const bool cost_row = !FULLEDGE || (t >= A.cr0 && t < A.cr1);
      const bool zero00 = FULLEDGE && t == 0 && b.C0 == 0;
      if (OUT) {
        mreg_row<T, S, REGK, R, C, RBD, true>(acc, st.cost_reg, xr, csrow, c2, lane, t, b.gc0, A.W, A.H, A.powtab, A.pwsum, cost_row, zero00, st.cmk);
      } else {
        T dacc[S];
        double dc = 0.0;
        mreg_row<T, S, REGK, R, C, RBD, false>(dacc, dc, xr, csrow, c2, lane, t, b.gc0, A.W, A.H, A.powtab, A.pwsum, false, zero00, st.cmk);
      }
    }
    // the left halo columns of this row, from the table of the prologue
    if (RU > 0 && outg && lane < RU) {
      const T hv = csh[(t - (b.R0 - RU)) * RU + lane];
      T* c0row = csrow - lane;  // lane 0's pointer
      if (lane == 0) c0row[mci<C>(-1)] = hv;
      else c0row[mci<C>(-2)] = hv;
    }
    m_pin(acc);
  }

#ifdef SRMAP_DEV_INSTANCES
  const unsigned long long ck2 = __builtin_readcyclecounter();
#endif
  // row t: data gradient, pass 2
  if (OUT && outg) {
    if (want_reg && RU > 0) {
      const T* xu[RU + 1];
      const T* cu[RU + 1];
#pragma unroll
      for (int i = 0; i <= RU; ++i) {
        xu[i] = xw[RU - i];
        cu[i] = cw[(NRC > 0 ? NRC - 1 : 0) - i];
      }
      mreg_pass2<T, S, REGK, R, C>(acc, xu, cu, A.powtab);
    }
    m_pin(acc);
  }

#ifdef SRMAP_DEV_INSTANCES
  const unsigned long long ck3 = __builtin_readcyclecounter();
#endif
  // ---------------- B: every request so far has landed ----------------
  // An explicit full drain the compiler accounts for: left to itself it waits here for the x row only and then, at the top
  // of the next iteration, for the weights with a count that also drains the stores and requests issued a moment before.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  m_put_x<T, S, C>(xs + st.ph * C::XROW, lane, st.nva, st.nvb, st.nma, st.nmb);  // x row t+HD+1 replaces row t-RU
  if (OUT && outg) {
    if (FULLEDGE) {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) acc[pc] -= cq[pc];
    }
    if (WD) {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) st.gd += (double)acc[pc] * (double)dreg[pc];
    }
    // ---------------- C: store g row t ----------------
    if (!SLOW || b.gc0 < A.W) {
      T* dst = A.g + (size_t)b.ch * b.N + (size_t)t * A.W + b.gc0;
#pragma unroll
      for (int pc = 0; pc < S; ++pc) __builtin_nontemporal_store(acc[pc], &dst[pc]);  // written once, not re-read here
    }
  }

  // ---------------- D: request x row t+HD+2 and the observations of residual row t+2+HB ----------------
  m_issue_x<T, S, C, DM>(A, b, t + HD + 2, st.nva, st.nvb, st.nma, st.nmb);
  if (want_data) m_issue_y<T, S, C, DM>(A, b, t + 2 + HB, PP ? st.yb : st.ya);  // into the buffer this row has consumed

#ifdef SRMAP_DEV_INSTANCES
  {
    const unsigned long long ck4 = __builtin_readcyclecounter();
    st.t_z += ck1 - ck0; st.t_pass1 += ck2 - ck1; st.t_p2 += ck3 - ck2; st.t_wait += ck4 - ck3;
  }
#endif
  st.ph = (st.ph + 1 == NRX) ? 0 : st.ph + 1;
  if (NRC > 0) st.phc = (st.phc + 1 == NRC) ? 0 : st.phc + 1;
}

// The band, from its first request to its last row: prologue (requests, left halo columns, ring fill), the PRE rows
// before the first output row (pass 1 values / residual rows only), the output rows.  Every code-path variant is its
// own instance of this function: nothing but the band description is live across the choice of the variant.
template <typename T, int S, int B, int REGK, int R, bool WD, int DM, bool RBD, bool SIMPLE, typename ArgsT>
__device__ __forceinline__ void march_band(const ArgsT& A, const MBand<T>& b, T* __restrict__ xs, T* __restrict__ cs,
                                           T* __restrict__ csh, double& cost_data, double& cost_reg, double& gd) {
  using C = MCfg<T, S, B, REGK, R>;
  constexpr int HB = C::HB, HD = C::HD, WIN = C::WIN, RU = C::RU, NV = C::NV;
  constexpr bool SLOW = DM != 0;
  const int lane = b.lane;
  const bool want_data = SIMPLE || b.want_data, want_reg = SIMPLE ? (REGK != 0) : b.want_reg, outg = SIMPLE || b.outg;
  MState<T, S, C> st;
  // ---- requests: the left halo columns' windows first (small, and their arithmetic runs while the rows are in
  // flight), then the band's first HD + 1 rows of x, the first row's weights and observations ----
  const bool halo_on = RU > 0 && want_reg && outg;
  MHalo<T, C> hreg;
  if (halo_on) march_halo_issue<T, S, REGK, R, C>(A, b.xplane, b.wplane, b.R0, b.tend, b.C0, lane, 0, hreg);
  T pva[HD + 1][S], pvb[HD + 1], pma[HD + 1], pmb[HD + 1];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) st.w[pc] = T(1);
#pragma unroll
  for (int v = 0; v < NV; ++v) { st.ya[v] = T(0); st.yb[v] = T(0); }
#pragma unroll
  for (int k = 0; k <= HD; ++k) m_issue_x<T, S, C, DM>(A, b, b.t0 + k, pva[k], pvb[k], pma[k], pmb[k]);
  if (want_reg) m_issue_w<T, S, DM, SIMPLE>(A, b, b.t0, st.w);
  if (want_data) m_issue_y<T, S, C, DM>(A, b, b.t0 + HB, st.ya);
  // ---- left halo columns of 2*lambda*w*r ----
  if (halo_on) {
    march_halo_finish<T, S, REGK, R, C>(A, csh, lane, 0, hreg);
    for (int base = 64; base < (b.tend - b.R0 + RU) * RU; base += 64) {  // tall bands only
      march_halo_issue<T, S, REGK, R, C>(A, b.xplane, b.wplane, b.R0, b.tend, b.C0, lane, base, hreg);
      march_halo_finish<T, S, REGK, R, C>(A, csh, lane, base, hreg);
    }
  }
  // ---- x rows -> ring ----
#pragma unroll
  for (int k = 0; k <= HD; ++k) m_put_x<T, S, C>(xs + (RU + k) * C::XROW, lane, pva[k], pvb[k], pma[k], pmb[k]);

#pragma unroll
  for (int pc = 0; pc < S; ++pc) st.mk[pc] = (!SLOW || b.gc0 + pc < A.W) ? T(1) : T(0);
#pragma unroll
  for (int c = 0; c < (WIN > 0 ? WIN : 1); ++c) st.cmk[c] = (!RBD || b.gc0 + S + c < A.W) ? T(1) : T(0);
#pragma unroll
  for (int pc = 0; pc < S; ++pc) { st.p1[pc] = T(0); st.p2[pc] = T(0); }
  st.ph = 0; st.phc = 0;
  st.cost_data = 0.0; st.cost_reg = 0.0; st.gd = 0.0;
#ifdef SRMAP_DEV_INSTANCES
  st.t_wait = 0; st.t_z = 0; st.t_pass1 = 0; st.t_p2 = 0;
#endif
  m_issue_x<T, S, C, DM>(A, b, b.t0 + HD + 1, st.nva, st.nvb, st.nma, st.nmb);  // "D" of a virtual iteration t0 - 1
  if (want_data) m_issue_y<T, S, C, DM>(A, b, b.t0 + 1 + HB, st.yb);
  // the PRE rows before the first output row, then the output rows two at a time (ping-pong observation buffers)
  constexpr int PRE = C::PRE;
  if (PRE >= 1) m_step<T, S, B, REGK, R, WD, DM, RBD, SIMPLE, false, 0>(A, b, xs, cs, csh, st, b.t0);
  if (PRE >= 2) m_step<T, S, B, REGK, R, WD, DM, RBD, SIMPLE, false, 1>(A, b, xs, cs, csh, st, b.t0 + 1);
  static_assert(PRE <= 2, "pre-rows are unrolled by hand");
  int t = b.R0;
#pragma unroll 1
  for (; t + 1 < b.tend; t += 2) {
    m_step<T, S, B, REGK, R, WD, DM, RBD, SIMPLE, true, (PRE & 1)>(A, b, xs, cs, csh, st, t);
    m_step<T, S, B, REGK, R, WD, DM, RBD, SIMPLE, true, (PRE & 1) ^ 1>(A, b, xs, cs, csh, st, t + 1);
  }
  if (t < b.tend) m_step<T, S, B, REGK, R, WD, DM, RBD, SIMPLE, true, (PRE & 1)>(A, b, xs, cs, csh, st, t);
  cost_data = st.cost_data; cost_reg = st.cost_reg; gd = st.gd;
#ifdef SRMAP_DEV_INSTANCES
  if (A.dbg != nullptr && lane == 0) {
    unsigned long long* dd = A.dbg + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8;
    dd[1] = st.t_wait; dd[6] = (st.t_z << 32) | (st.t_pass1 & 0xffffffffull); dd[7] = (st.t_p2 << 32) | (dd[7] & 0xffffffffull);
  }
#endif
}

// SIMPLE (host-decided): data term + fused regulariser + gradient requested, IRLS weights present, one residual per
// pixel phase -- the loop bodies are straight-line code.  Otherwise the general loop bodies.
template <typename T, int S, int B, int REGK, int R, bool WD, bool SIMPLE>
__global__ __launch_bounds__(64, 2) void k_eval_march(MArgs<T, B, MCfg<T, S, B, REGK, R>::NP> A) {
  using C = MCfg<T, S, B, REGK, R>;
  constexpr int HB = C::HB, NV = C::NV, RU = C::RU, WIN = C::WIN, HD = C::HD;
  constexpr int PRE = C::PRE;
  constexpr int RING_ELEMS = C::XS_ELEMS + C::CS_ELEMS + C::CSH_ELEMS;
  constexpr int BORDER_ELEMS = (kMarchBorderLds + (int)sizeof(T) - 1) / (int)sizeof(T);
  __shared__ T lds[RING_ELEMS > BORDER_ELEMS ? RING_ELEMS : BORDER_ELEMS];  // ONE LDS object (wave-private)
  T* const xs = lds;
  T* const cs = lds + C::XS_ELEMS;
  T* const csh = cs + C::CS_ELEMS;

  const int lane = threadIdx.x;
  const int gw = blockIdx.y * gridDim.x + blockIdx.x;  // wave index in dispatch order
#ifdef SRMAP_DEV_INSTANCES
  unsigned long long ts0 = __builtin_amdgcn_s_memrealtime(), ts1 = 0, ts2 = 0, ts3 = 0;
#endif
  // XCD-aware order: workgroups are dealt to the 8 XCDs round robin in launch order and every XCD has its own L2.
  // Launch index n -> work item m such that an XCD runs a contiguous run of (strip, band) pairs, band fastest:
  // vertical neighbours (they share RU + HD of their x rows) run on the same XCD at the same time.
  int m;
  {
    const int n = blockIdx.x, q = gridDim.x >> 3, rem = gridDim.x & 7, bnd = n & 7;
    m = bnd * q + (bnd < rem ? bnd : rem) + (n >> 3);
  }
  int strip, band, RB;
  if (m < A.nbe || A.nstrips <= 2) { strip = m / A.nbe; band = m - strip * A.nbe; RB = A.RBe; }
  else {
    const int mm = m - A.nbe, ni = (A.nstrips - 2) * A.nbands;
    if (mm < ni) { const int q = mm / A.nbands; strip = 1 + q; band = mm - q * A.nbands; RB = A.RB; }
    else { strip = A.nstrips - 1; band = mm - ni; RB = A.RBe; }
  }
  MBand<T> b;
  b.lane = lane;
  b.ch = blockIdx.y;
  b.R0 = band * RB;
  b.tend = (b.R0 + RB < A.H) ? b.R0 + RB : A.H;
  b.CJ0 = strip * C::CW;
  b.C0 = b.CJ0 * S;
  b.gc0 = b.C0 + S * lane;  // first global HR column of this thread
  b.t0 = b.R0 - PRE;
  b.N = (size_t)A.W * A.H;
  const size_t nl = (size_t)A.wl * A.hl;
  b.xplane = A.x + (size_t)b.ch * b.N;
  b.ybase = A.y + (size_t)b.ch * nl;
  b.want_data = (A.terms & SRMAP_TERM_DATA) != 0;
  b.want_reg = REGK != 0 && (A.terms & SRMAP_TERM_REG) != 0;
  b.outg = A.g != nullptr;
  b.wplane = (b.want_reg && A.w) ? A.w + (size_t)b.ch * b.N : nullptr;
  const int R0 = b.R0, CJ0 = b.CJ0, C0 = b.C0;
  // bands whose residuals can touch LR row / column 0 or leave the LR image take the masked path (uniform); so do
  // row-band problems (cost rows restricted), partial bands / strips and bands whose windows or prefetch rows leave
  // the image
  const int rm = A.E + HB + 1, cm = (A.E + HB + S) / S + 1;
  const bool edge = (R0 - rm < 0) || (R0 + RB + rm > A.H) || (CJ0 - cm < 0) || (CJ0 + C::CW + cm > A.wl) ||
                    A.cr0 > 0 || A.cr1 < A.H;
  const bool reg_border = (R0 + RB + WIN > A.H) || (C0 + C::TW + WIN > A.W);
  const bool slow = edge || reg_border || R0 < PRE || R0 + RB + HD + 2 > A.H || R0 + RB + rm + 2 > A.H ||
                    CJ0 < C::XCL || CJ0 + C::CW + C::XCR > A.wl;
  // in-image pixels of the border frame (gradient corrections from the border tasks): top rows / left columns only
  b.has_ring = b.want_data && b.outg && (R0 < A.ring.rg[0] || C0 < A.ring.rg[1]);
  b.corr = b.has_ring ? A.bd->corr + (size_t)b.ch * A.bd->n_ring : nullptr;

  // data-path mode of this band: 0 interior; 1 only columns can leave the image (SIMPLE kernels); 2 general
  const bool rows_in = !(R0 - rm < 0) && !(R0 + RB + rm + 2 > A.H) && A.cr0 == 0 && A.cr1 >= A.H && R0 >= PRE &&
                       R0 + RB + HD + 2 <= A.H && R0 + RB + WIN <= A.H;
  const int dm = !slow ? 0 : ((SIMPLE && rows_in && !b.has_ring && A.ring.rg[1] == 0) ? 1 : 2);

  // (border tasks: when some wave consumes their corrections they ran in k_march_border, launched before this kernel;
  // when they only produce cost partials they are picked up at the end of this kernel by the waves that finish first)
#ifdef SRMAP_DEV_INSTANCES
  ts1 = __builtin_amdgcn_s_memrealtime();
  ts2 = ts1;
#endif
  double cost_data = 0.0, cost_reg = 0.0, gd = 0.0;
  // three code paths (each variant more in one kernel costs every variant registers): interior; first / last strips away
  // from the image top / bottom (column masks only; the regulariser's right-edge masks ride along); everything else
  if (dm == 0) march_band<T, S, B, REGK, R, WD, 0, false, SIMPLE>(A, b, xs, cs, csh, cost_data, cost_reg, gd);
  else if (SIMPLE && dm == 1) march_band<T, S, B, REGK, R, WD, (SIMPLE ? 1 : 2), true, SIMPLE>(A, b, xs, cs, csh, cost_data, cost_reg, gd);
  else march_band<T, S, B, REGK, R, WD, 2, true, SIMPLE>(A, b, xs, cs, csh, cost_data, cost_reg, gd);

#ifdef SRMAP_DEV_INSTANCES
  ts3 = __builtin_amdgcn_s_memrealtime();
#endif
  // ---------------- cost partial of this wave; finish ----------------
  const double cwv = wave_sum_d((double)(S * S) * cost_data + cost_reg);
  double gdv = 0.0;
  if (WD) gdv = wave_sum_d(gd);
  const unsigned nwaves = gridDim.x * gridDim.y;
#ifdef SRMAP_DEV_INSTANCES
  if (A.dbg != nullptr && lane == 0) {
    unsigned long long* d = A.dbg + (size_t)gw * 8;
    d[0] = ts0; d[2] = ts2; d[3] = ts3; d[4] = __builtin_amdgcn_s_memrealtime();
    d[5] = (unsigned long long)slow | ((unsigned long long)0 << 1) | ((unsigned long long)b.has_ring << 2) | ((unsigned long long)reg_border << 3) | ((unsigned long long)dm << 4);
    d[5] |= ((unsigned long long)strip << 40) | ((unsigned long long)band << 16);
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    d[7] = (d[7] & 0xffffffff00000000ull) | (hw & 0xfffffu) | ((unsigned long long)xcc << 24);
  }
#endif
  if (A.finish) {
    // Every wave publishes its partials as write-through granules and leaves; the LAST wave of the grid (dispatched last,
    // among the last to finish) gathers them: it polls until no granule holds the sentinel, adds them in index order
    // (deterministic), re-arms the granules and publishes the cost.  No ticket, no wait on this wave's own stores.
    if (lane == 0) {
      st_agent(&A.mpart[gw], cwv);
      if (WD) st_agent(&A.mpart_gd[gw], gdv);
    }
    if ((unsigned)gw != nwaves - 1) {
      // border tasks nobody waits for: picked up by the waves that finish first
      if (A.duty_at_end && b.want_data) {
        bool tables = false;
        while (true) {
          unsigned long long tkn = 0;
          if (lane == 0) tkn = __hip_atomic_fetch_add(A.ctr64, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - A.task_base;
          const unsigned tlo = __builtin_amdgcn_readfirstlane((unsigned)tkn), thi = __builtin_amdgcn_readfirstlane((unsigned)(tkn >> 32));
          if (thi != 0 || tlo >= (unsigned)A.ntasks_total) break;
          if (!tables) { march_border_tables<T, S>(*A.bd, lane, (void*)lds); tables = true; }
          march_border_task<T, S, B>(A, *A.bd, (int)tlo, lane, (void*)lds);
        }
      }
      return;
    }
    constexpr int U = 40;  // requests in flight per lane and round; the order of the additions is fixed
    double v = 0.0, v2 = 0.0;
    bool timed_out = false;
    for (int pass = 0; pass < (WD ? 2 : 1); ++pass) {
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(pass == 0 ? A.mpart : A.mpart_gd);
      const int n = pass == 0 ? A.n_partials : A.n_wave_partials;
      double acc = 0.0;
      for (int base = 0; base < n; base += 64 * U) {
        unsigned long long a[U];
        unsigned spins = 0;
        while (true) {
          bool missing = false;
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int i = base + u * 64 + lane;
            a[u] = ld_agent(&src[i < n ? i : base]);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int i = base + u * 64 + lane;
            missing |= (i < n) && a[u] == kSentinel;
          }
          if (!__any(missing)) break;
          if (++spins > (1u << 14)) { timed_out = true; break; }
          __builtin_amdgcn_s_sleep(4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = base + u * 64 + lane;
          acc += (i < n) ? __longlong_as_double((long long)a[u]) : 0.0;
        }
      }
      acc = wave_sum_d(acc);
      if (pass == 0) v = acc; else v2 = acc;
    }
    // re-arm: the next evaluation finds every granule unpublished
    for (int i = lane; i < A.n_partials; i += 64) st_agent(reinterpret_cast<unsigned long long*>(A.mpart) + i, kSentinel);
    if (WD)
      for (int i = lane; i < A.n_wave_partials; i += 64) st_agent(reinterpret_cast<unsigned long long*>(A.mpart_gd) + i, kSentinel);
    if (lane == 0) {
      const unsigned bad = ld_agent(&A.ctr[2]);
      if (bad || timed_out) v = __builtin_nan("");  // a wait timed out: the evaluation is not trustworthy
      A.cost_out[0] = v;
      if (WD) {
        A.cost_out[1] = v2;
        if (A.pub != nullptr) {  // solver line search: {cost, g.d} straight to the host-mapped words, then the arrival tag
          A.pub[0] = v;
          A.pub[1] = v2;
          __threadfence_system();
          *(volatile double*)A.tag_slot = A.tag;
        }
      }
      st_agent(&A.ctr[1], 0u);
      st_agent(&A.ctr[2], 0u);
    }
    return;
  }
  // the caller reduces the partials (more kernels follow, or too many partials): plain stores; the last arriver of a
  // ticket re-arms the border counter
  unsigned tk = 0;
  if (lane == 0) {
    A.partials[gw] = cwv;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tk = __hip_atomic_fetch_add(&A.ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tk == nwaves - 1) {
      st_agent(&A.ctr[0], 0u);
      st_agent(&A.ctr[1], 0u);
      st_agent(&A.ctr[2], 0u);
    }
  }
}

// Border tasks as their own launch (one wave per task), BEFORE k_eval_march: used when frames with positive offsets
// exist, i.e. when the corrections of the tasks are consumed by the bands (the kernel boundary orders them).
template <typename T, int S, int B, int NP>
__global__ __launch_bounds__(64) void k_march_border(MArgs<T, B, NP> A) {
  __shared__ T lds[(kMarchBorderLds + (int)sizeof(T) - 1) / (int)sizeof(T)];
  const int lane = threadIdx.x;
  march_border_tables<T, S>(*A.bd, lane, (void*)lds);
  march_border_task<T, S, B>(A, *A.bd, (int)blockIdx.x, lane, (void*)lds);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// host side
static int g_march_end_duty = 1;  // development builds can switch the end-of-wave pick-up of border tasks off
#ifdef SRMAP_DEV_INSTANCES
extern "C" void srmap_dev_set_end_duty(int v) { g_march_end_duty = v; }
static unsigned long long* g_march_dbg = nullptr;
extern "C" void srmap_dev_set_march_dbg(void* p) { g_march_dbg = (unsigned long long*)p; }
#endif
template <typename T, int S, int B, int REGK, int R>
static int launch_m(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g, const T* wts,
                    const ZPlan& z, double* partials, int* nblocks, bool finish_ok, bool* finished, hipStream_t st,
                    const T* dvec, double* partials_gd, bool publish) {
  using C = MCfg<T, S, B, REGK, R>;
  MArgs<T, B, C::NP> A;
  A.x = x; A.y = (const T*)p->d_obs + (size_t)obs_c0 * geo.w * geo.h; A.w = wts; A.g = g; A.partials = partials;
  A.cnt = z.d_cnt; A.off = z.d_off; A.aux = z.d_aux; A.MS = z.MS;
  for (int pr = 0; pr < 4; ++pr) {
    for (int i = 0; i < 8; ++i) A.cntk[pr][i] = z.h_cnt[pr * 8 + i];
    for (int pc = 0; pc < 4; ++pc) { A.off0[pr][pc] = z.h_off0[pr * 4 + pc]; A.aux0[pr][pc] = z.h_aux0[pr * 4 + pc]; }
  }
  A.W = geo.W; A.H = geo.H; A.wl = geo.w; A.hl = geo.h;
  A.obs_C = p->geo.C;
  A.E = z.E;
  A.ring = z.ring;
  A.cr0 = geo.cr0; A.cr1 = geo.cr1;
  A.rr0 = 0; A.rr1 = geo.H;
  A.terms = (int)terms;
  if (B == 1) { A.blur3[0] = A.blur3[1] = A.blur3[2] = T(1); A.k1s[0] = A.k1s[1] = T(1); }
  else {
    const int hb = (B - 1) / 2;
    A.blur3[0] = (T)p->blur2d[0]; A.blur3[1] = (T)p->blur2d[hb]; A.blur3[2] = (T)p->blur2d[hb * B + hb];
    A.k1s[0] = (T)p->blur1d[0]; A.k1s[1] = (T)p->blur1d[hb];
  }
  A.lambda = T(0);
  for (int i = 0; i < C::NP; ++i) A.powtab[i] = T(1);
  if (REGK != 0 && (terms & SRMAP_TERM_REG) && z.reg_index >= 0) {
    const RegSpec& rs = p->reg[z.reg_index];
    A.lambda = (T)rs.lambda;
    if (REGK == 2) for (int i = 0; i < C::NP; ++i) A.powtab[i] = (T)rs.pow_table[i];
  }
  A.pwsum = T(0);
  if (REGK == 2)
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < R; ++j)
        if (i + j > 0) A.pwsum += A.powtab[i + j];
  A.bd = (const BorderArgs<T>*)z.d_bd;
  A.nby = 0; A.n_tile_partials = 0;
  A.rbuf = nullptr; A.spw = nullptr; A.Dr = 0;
  // ---- bands: as few generations of waves as possible, each as short as possible ----
  bool simple = REGK != 0 && (terms & SRMAP_TERM_DATA) && (terms & SRMAP_TERM_REG) && g != nullptr && wts != nullptr;
  for (int pr = 0; pr < S; ++pr)
    for (int pc = 0; pc < S; ++pc)
      if (z.h_cnt[pr * 8 + pc] != 1) simple = false;
  const bool wd = dvec != nullptr && finish_ok;
  const void* kfn = simple ? (wd ? reinterpret_cast<const void*>(&k_eval_march<T, S, B, REGK, R, true, true>)
                                 : reinterpret_cast<const void*>(&k_eval_march<T, S, B, REGK, R, false, true>))
                           : (wd ? reinterpret_cast<const void*>(&k_eval_march<T, S, B, REGK, R, true, false>)
                                 : reinterpret_cast<const void*>(&k_eval_march<T, S, B, REGK, R, false, false>));
  static int occ_cache[4] = {0, 0, 0, 0};  // per kernel instance (this function is one template instance)
  int& occ = occ_cache[(wd ? 1 : 0) + (simple ? 2 : 0)];
  if (occ == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, 64, 0) != hipSuccess || occ < 1)) occ = 4;
  const int cus = p->ctx->num_cus > 0 ? p->ctx->num_cus : 256;
  const long long slots = (long long)cus * occ;
  const int nstrips = (geo.w + C::CW - 1) / C::CW;
  // band heights: one for the interior strips, one for the first / last strip (masked code paths, ~kSlow x the work per
  // row) -- as few generations of waves as possible, all waves of a generation ending together.  Cached per geometry.
  static long long rb_key[4] = {-1, -1, -1, -1};
  static int rb_val[2] = {8, 8};
  if (rb_key[0] != geo.H || rb_key[1] != geo.w || rb_key[2] != geo.C || rb_key[3] != slots) {
    const double kSlow = 1.3;
    double best = 1e300;
    int bi = 8, be = 8;
    for (int rbi = kMarchMinRB; rbi <= kMarchMaxRB; ++rbi) {
      const long long nbi = (geo.H + rbi - 1) / rbi;
      const double life_i = rbi + C::PRE * 0.5 + 1.0;
      for (int rbe = kMarchMinRB; rbe <= rbi; ++rbe) {
        const long long nbe = (geo.H + rbe - 1) / rbe;
        const double life_e = kSlow * (rbe + C::PRE * 0.5 + 1.0);
        long long wi, we;
        if (nstrips <= 2) { if (rbe != rbi) continue; wi = 0; we = nstrips * nbe * geo.C; }
        else { wi = (long long)(nstrips - 2) * nbi * geo.C; we = 2 * nbe * geo.C; }
        const double lmax = std::max(wi > 0 ? life_i : 0.0, life_e);
        double c;
        if (wi + we <= slots) c = lmax;                                                   // one generation
        else c = ((double)wi * life_i + (double)we * life_e) / (double)slots + 0.5 * lmax;  // throughput + tail
        if (c < best - 1e-9) { best = c; bi = rbi; be = rbe; }
      }
    }
    rb_key[0] = geo.H; rb_key[1] = geo.w; rb_key[2] = geo.C; rb_key[3] = slots;
    rb_val[0] = bi; rb_val[1] = be;
  }
  A.RB = rb_val[0];
  A.RBe = rb_val[1];
  A.nbands = (geo.H + A.RB - 1) / A.RB;
  A.nbe = (geo.H + A.RBe - 1) / A.RBe;
  A.nstrips = nstrips;
  dim3 grid((unsigned)(nstrips <= 2 ? nstrips * A.nbe : 2 * A.nbe + (nstrips - 2) * A.nbands), (unsigned)geo.C, 1);
  const int nwaves = (int)(grid.x * grid.y);
  A.ntasks = 0; A.ntasks_total = 0; A.nduty = 0;
  if ((terms & SRMAP_TERM_DATA) && z.n_ring > 0) {
    A.ntasks = (z.n_ring + 63) / 64;
    A.ntasks_total = A.ntasks * geo.C;
    // duty waves must all be resident before any wave waits for them: the head of the first generation
    const long long cap = std::max<long long>(1, std::min<long long>(slots / 2, nwaves));
    A.nduty = (int)std::min<long long>(A.ntasks_total, cap);
  }
  A.one_round = simple ? 1 : 0;
  A.n_wave_partials = nwaves;
  A.n_partials = nwaves + A.ntasks_total;
  const bool finish = finish_ok && z.d_mpart != nullptr && (size_t)A.n_partials <= z.mpart_cap;  // else the caller reduces the partials (two stages)
  if (!finish) dvec = nullptr;                             // g.d rides on the in-kernel reduction only
  A.dvec = dvec; A.partials_gd = partials_gd;
  A.finish = finish ? 1 : 0;
  *finished = finish;
  A.duty_at_end = 0; A.task_base = 0;
  if (g_march_end_duty && finish && A.ntasks_total > 0 && z.ring.rg[0] == 0 && z.ring.rg[1] == 0 && z.d_ctr64 != nullptr) {
    // no in-image correction pixels: nobody waits for the border tasks
    A.duty_at_end = 1;
    A.nduty = 0;
    A.task_base = z.task_count;
    z.task_count += (unsigned long long)A.ntasks_total + (unsigned long long)(nwaves - 1);  // every picker ends on a miss
  }
  A.ctr = z.d_ctr;
  A.cost_out = p->d_cost;
  A.pub = (publish && dvec != nullptr) ? p->eval_pub : nullptr;
  A.tag_slot = p->eval_pub_tag_slot;
  A.tag = p->eval_pub_tag;
  A.ctr64 = z.d_ctr64;
  A.mpart = z.d_mpart;
  A.mpart_gd = z.d_mpart ? z.d_mpart + z.mpart_cap : nullptr;
  A.dbg = nullptr;
#ifdef SRMAP_DEV_INSTANCES
  A.dbg = g_march_dbg;
#endif
  if (A.ntasks_total > 0 && !A.duty_at_end)
    hipLaunchKernelGGL((k_march_border<T, S, B, C::NP>), dim3((unsigned)A.ntasks_total), dim3(64), 0, st, A);
  if (simple) {
    if (dvec != nullptr) hipLaunchKernelGGL((k_eval_march<T, S, B, REGK, R, true, true>), grid, dim3(64), 0, st, A);
    else hipLaunchKernelGGL((k_eval_march<T, S, B, REGK, R, false, true>), grid, dim3(64), 0, st, A);
  } else {
    if (dvec != nullptr) hipLaunchKernelGGL((k_eval_march<T, S, B, REGK, R, true, false>), grid, dim3(64), 0, st, A);
    else hipLaunchKernelGGL((k_eval_march<T, S, B, REGK, R, false, false>), grid, dim3(64), 0, st, A);
  }
  *nblocks = A.n_partials;
  SRMAP_HIP(p->ctx, hipGetLastError());
  return SRMAP_OK;
}

// The instances compiled: the marching kernel is an opt-in implementation (SRMAP_IMPL_MARCH), built for the geometries of
// the BASELINE configurations -- (scale, blur size, fused regulariser kind, BTV range).
#ifdef SRMAP_DEV_INSTANCES
#define SRMAP_MARCH_INSTANCES(X) X(4, 3, 2, 3)
#else
#define SRMAP_MARCH_INSTANCES(X) X(4, 3, 2, 3) X(4, 1, 2, 3) X(3, 1, 1, 0) X(2, 1, 1, 0)
#endif

bool march_has_instance(int S, int B, int regk, int regr) {
#define X(s_, b_, k_, r_) if (S == s_ && B == b_ && (regk == 0 || (regk == k_ && regr == r_))) return true;
  SRMAP_MARCH_INSTANCES(X)
#undef X
  return false;
}

// One launch: data term + the fused regulariser + border corrections (+ the cost reduction when `finish`).
// *nblocks = partials written (the caller reduces them when !*finished).
template <typename T>
int launch_eval_march(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g, const T* wts,
                      int regk, int regr, double* partials, int* nblocks, bool finish, bool* fin, hipStream_t st,
                      const T* dv, double* pgd, bool publish) {
  const ZPlan& z = *static_cast<const ZPlan*>(p->zplan);
  const int S = geo.s, B = geo.b;
  // regk == 0 (the fused regulariser is not part of this evaluation): any instance of the geometry serves, its
  // regulariser switched off by `terms`
#define X(s_, b_, k_, r_)                                                                                              \
  if (S == s_ && B == b_ && (regk == 0 || (regk == k_ && regr == r_)))                                                  \
    return launch_m<T, s_, b_, k_, r_>(p, geo, obs_c0, terms, x, g, wts, z, partials, nblocks, finish, fin, st, dv, pgd, publish);
  SRMAP_MARCH_INSTANCES(X)
#undef X
  return set_error(p->ctx, SRMAP_EUNSUPPORTED, "no marching kernel for scale %d blur %d regulariser %d/%d", S, B, regk, regr);
}

template int launch_eval_march<float>(srmap_problem*, const Geometry&, int, unsigned, const float*, float*, const float*,
                                      int, int, double*, int*, bool, bool*, hipStream_t, const float*, double*, bool);
template int launch_eval_march<double>(srmap_problem*, const Geometry&, int, unsigned, const double*, double*,
                                       const double*, int, int, double*, int*, bool, bool*, hipStream_t, const double*,
                                       double*, bool);

// HIP loads a kernel's code object lazily at its first launch (milliseconds): touch the instances when the plan is made.
template <typename T, int S, int B, int REGK, int R>
static void preload_m() {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_eval_march<T, S, B, REGK, R, false, false>));
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_eval_march<T, S, B, REGK, R, true, false>));
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_eval_march<T, S, B, REGK, R, false, true>));
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_eval_march<T, S, B, REGK, R, true, true>));
}
void march_preload(const srmap_problem* p) {
  const ZPlan* z = static_cast<const ZPlan*>(p->zplan);
  if (!z || z->subpix || p->impl != SRMAP_IMPL_MARCH) return;
#define X(s_, b_, k_, r_)                                                             \
  if (z->S == s_ && z->B == b_ && z->regk == k_ && (k_ != 2 || z->regr == r_)) {       \
    if (p->dtype == SRMAP_F32) preload_m<float, s_, b_, k_, r_>();                     \
    else preload_m<double, s_, b_, k_, r_>();                                          \
  }
  SRMAP_MARCH_INSTANCES(X)
#undef X
}

bool march_alloc(srmap_problem* p, ZPlan* z) {
  if (hipMalloc((void**)&z->d_ctr, 4 * sizeof(unsigned)) != hipSuccess) return false;
  if (hipMemset(z->d_ctr, 0, 4 * sizeof(unsigned)) != hipSuccess) return false;
  if (hipMalloc((void**)&z->d_ctr64, sizeof(unsigned long long)) != hipSuccess) return false;
  if (hipMemset(z->d_ctr64, 0, sizeof(unsigned long long)) != hipSuccess) return false;
  z->task_count = 0;
  // granules for the in-kernel reduction: enough for every band height the launcher may choose, up to a cap beyond
  // which the caller's two-stage reduction is used anyway
  const size_t cap = std::min<size_t>(march_partials_needed(p), (size_t)16384);
  if (hipMalloc((void**)&z->d_mpart, 2 * cap * sizeof(double)) != hipSuccess) return false;
  if (hipMemsetD32((hipDeviceptr_t)z->d_mpart, (int)kSentinel32, 4 * cap) != hipSuccess) return false;
  z->mpart_cap = cap;
  return true;
}

size_t march_partials_needed(const srmap_problem* p) {
  const Geometry& g = p->geo;
  const ZPlan* z = static_cast<const ZPlan*>(p->zplan);
  const size_t waves = (size_t)((g.w + 63) / 64) * ((g.H + kMarchMinRB - 1) / kMarchMinRB) * g.C;
  const size_t tasks = (z && z->n_ring > 0) ? (size_t)((z->n_ring + 63) / 64) * g.C : 0;
  return waves + tasks;
}

}  // namespace srmap
