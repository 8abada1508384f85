// ztile_dev.hpp -- device helpers, kernel-argument block and the host-side plan of the hot-path kernel
// (kernels_ztile.hip: workgroup tiles).  Formulation: DESIGN.md section 3.1.
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "cg_norm.hpp"
#include "srmap_internal.hpp"

namespace srmap {

// integer floor division / modulo on the host
static inline int fdiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
static inline int pmod(int a, int b) { return a - fdiv(a, b) * b; }

namespace {

// Sum over the 64 lanes of a wave, result in EVERY lane.  DPP lane permutations (register-to-register: the
// __shfl_down tree went through ds_bpermute, twelve dependent LDS round trips at the end of every workgroup's life)
// inside the rows of 16 lanes, then the four row sums through v_readlane.  Fixed order: deterministic.
template <int CTRL>
__device__ __forceinline__ double dpp_perm_d(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum_d(double v) {
  v += dpp_perm_d<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_perm_d<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_perm_d<0x141>(v);  // row_half_mirror
  v += dpp_perm_d<0x140>(v);  // row_mirror: every lane holds the sum of its row of 16
  return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}

constexpr int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
constexpr int posmod(int a, int b) { return a - floordiv(a, b) * b; }

template <typename T>
__device__ __forceinline__ T absv(T d) { return d < T(0) ? -d : d; }
template <>
__device__ __forceinline__ float absv<float>(float d) { return __builtin_fabsf(d); }
template <>
__device__ __forceinline__ double absv<double>(double d) { return __builtin_fabs(d); }

// Pin a set of accumulators at this point of the program: what was computed into them so far is complete here and no memory
// access moves across.  Left alone the scheduler gathers the LDS reads of EVERY window row of a regulariser pass at its head
// and the register allocation follows the window data in flight, not the arithmetic: with the two regulariser passes pinned
// per row (round 6) the blur-free f64 instances drop from 116 to 72 VGPRs -- 6 waves per SIMD = three workgroups per CU at
// their 53 KB of LDS -- and cfg3 runs 0.394 -> 0.358 ms, cfg5 77.5 -> 72.6 us per channel, f32 26.3 -> 25.6 us; the cfg2
// instance (124 -> 126 VGPRs, 73.5 KB) is unchanged.  Same instructions, same results bit for bit.
// SRMAP_EXP_PIN = bit mask for A/B builds: 1 = regulariser pass 1 (per window row), 2 = regulariser pass 2 (per neighbour
// row), 4 = data term (per blur row: costs the cfg2 instance spills -- off).  0 = the schedule of rounds 3 - 5.
#ifndef SRMAP_EXP_PIN
#define SRMAP_EXP_PIN 3
#endif
template <int WHICH, typename T, int N>
__device__ __forceinline__ void pin(T (&a)[N]) {
  if ((SRMAP_EXP_PIN & WHICH) != 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(a[i]) : : "memory");
  }
}

constexpr int zmax(int a, int b) { return a > b ? a : b; }
constexpr int zceil(int a, int b) { return (a + b - 1) / b; }

#ifndef SRMAP_ZT_NW
#define SRMAP_ZT_NW 8   // measurement builds: waves (= HR rows) per tile workgroup
#endif
template <typename T, int S, int B, int REGK, int R, int NW_ = SRMAP_ZT_NW>
struct ZCfg {
  static constexpr int NW = NW_;               // waves = HR rows per tile
  static constexpr int NT = 64 * NW;
  static constexpr int TH = NW;
  static constexpr int CW = 64;                // LR cells per tile row = lanes
  static constexpr int TW = CW * S;
  static constexpr int HB = (B - 1) / 2;
  static constexpr int WIN = REGK == 2 ? R : (REGK == 1 ? 1 : 0);      // pass 1 reaches WIN pixels right / down
  static constexpr int RU = REGK == 2 ? R - 1 : (REGK == 1 ? 1 : 0);   // pass 2 reaches RU pixels up / left
  static constexpr int HU = zmax(RU, 2 * HB);  // x halo rows above / below the tile
  static constexpr int HD = zmax(WIN, 2 * HB);
  static constexpr int XCL = zceil(zmax(RU, 2 * HB), S);   // x halo cells left / right
  static constexpr int XCR = zceil(zmax(WIN, 2 * HB), S);
  static constexpr int XC = CW + XCL + XCR;
  static constexpr int XR = TH + HU + HD;
  static constexpr int XROW = S * XC;
  static constexpr int XS_ELEMS = XR * XROW;
  static constexpr int NV = S + 2 * HB;        // pixels a thread evaluates B x / z at: own S + HB each side
  static constexpr int ZR = (B > 1) ? TH + 2 * HB : 0;     // zh rows -HB .. TH-1+HB (own columns only)
  static constexpr int ZROW = S * CW;
  static constexpr int ZS_ELEMS = ZR * ZROW;
  static constexpr int CCL = RU > 0 ? zceil(RU, S) : 0;    // 2*lambda*w*r: halo cells on the left
  static constexpr int CC = CW + CCL;
  static constexpr int CROW = S * CC;
  static constexpr int CRR = REGK ? TH + RU : 0;
  static constexpr int CS_ELEMS = CRR * CROW;
  static constexpr int NP = REGK == 2 ? 2 * R + 1 : 1;
};

// Sub-pixel instances, source-major table: one record per (frame, vertical tap of the transpose warp) that lands on a
// row phase -- LR row offset io, the two horizontal tap weights (already multiplied by the vertical one) and what follows
// from the frame's integer column offset ox alone, worked out on the host (the scalar unit of a CU is shared by its
// sixteen waves: per-source divisions and range tests there were a fifth of the kernel): a = (-ox) mod S, the residual
// column offset q = (ox + a) / S of e = 0, and the first of the residual columns e in {-1, 0, 1} a cell's S + 2 HB pixels
// use (am = a | (e_lo + 1) << 8; the kSpSlots(S, HB) columns from e_lo on are requested unconditionally -- no per-column
// test on the scalar unit).  Rows of the table are padded to a multiple of kSpChunk with null records (am = kSpNull,
// weights 0): a chunk's records are loaded back to back without control flow in between (z_row_sp2).
struct ZSrc { int k, io, q, am; double w0, w1; };
constexpr int kSpNull = 0xff;
__host__ __device__ constexpr int kSpSlots(int S, int HB) { return (S == 2 && HB == 1) ? 3 : 2; }
struct ZEntry { int k, io, jo, oyx; };  // frame, LR row / column offset of the residual a pixel of this phase owns,
                                         // forward offset packed (oy << 16) | (ox & 0xffff)

// The pixels of the border frame that can carry work, as six rectangles in HR coordinates (host: ring_rects()).
// With blur reach (B - 1) / 2 <= 1 < S a frame k with forward offset o_k = (ox, oy) has
//   * a residual whose z position S (i, j) + o_k lies OUTSIDE the image only in the rows [min oy, 0) / [H, H + max oy - S]
//     and the columns [min ox, 0) / [W, W + max ox - S]                                (cost of ownerless residuals),
//   * a contribution to an in-image pixel q that the clipped transpose warp excludes (q - o_k outside, while a residual
//     sits within blur reach of q) only for q - o_k = -1 in a row or column, i.e. rows [0, max oy) and columns [0, max ox)
//     (gradient corrections).
// Everything else of the 2E-wide frame around the image edge evaluates to zero and is not enumerated.
//   rg[0] = ti: TOP-INSIDE     rows [0, ti)       x columns [0, W)
//   rg[1] = li: LEFT-INSIDE    columns [0, li)    x rows [ti, H)
//   rg[2] = to: TOP-OUTSIDE    rows [-to, 0)      x columns [-lo, W + ro)
//   rg[3] = bo: BOTTOM-OUTSIDE rows [H, H + bo)   x columns [-lo, W + ro)
//   rg[4] = lo: LEFT-OUTSIDE   columns [-lo, 0)   x rows [0, H)
//   rg[5] = ro: RIGHT-OUTSIDE  columns [W, W+ro)  x rows [0, H)
struct RingRects { int rg[6]; };

__host__ __device__ __forceinline__ long long ring_count(const RingRects& r, int W, int H) {
  const long long ti = r.rg[0], li = r.rg[1], to = r.rg[2], bo = r.rg[3], lo = r.rg[4], ro = r.rg[5];
  return ti * W + li * (H - ti) + (to + bo) * (W + lo + ro) + (lo + ro) * H;
}

// pixel of the frame for thread index t
__device__ __forceinline__ void ring_pixel(int t, int W, int H, const RingRects& r, int& qr, int& qc) {
  const int ti = r.rg[0], li = r.rg[1], to = r.rg[2], bo = r.rg[3], lo = r.rg[4], ro = r.rg[5];
  const int n0 = ti * W;
  if (t < n0) { qr = t / W; qc = t - qr * W; return; }
  t -= n0;
  const int n1 = li * (H - ti);
  if (t < n1) { const int u = t / li; qr = ti + u; qc = t - u * li; return; }
  t -= n1;
  const int We = W + lo + ro;
  const int n2 = to * We;
  if (t < n2) { const int u = t / We; qr = -to + u; qc = -lo + (t - u * We); return; }
  t -= n2;
  const int n3 = bo * We;
  if (t < n3) { const int u = t / We; qr = H + u; qc = -lo + (t - u * We); return; }
  t -= n3;
  const int n4 = lo * H;
  if (t < n4) { qr = t / lo; qc = -lo + (t - qr * lo); return; }
  t -= n4;
  qr = t / ro; qc = W + (t - qr * ro);
}
// index of the IN-IMAGE frame pixel (qr, qc) in the correction array (inverse of ring_pixel); -1: not on the frame
__device__ __forceinline__ int ring_index(int qr, int qc, int W, int H, const RingRects& r) {
  const int ti = r.rg[0], li = r.rg[1];
  if (qr < ti) return qr * W + qc;
  if (qc < li) return ti * W + (qr - ti) * li + qc;
  return -1;
}

// A granule that has not been published yet holds this NaN pattern (both 32-bit halves equal: hipMemsetD32 writes it).
constexpr unsigned kSentinel32 = 0x7FF9ABCDu;
constexpr unsigned long long kSentinel = ((unsigned long long)kSentinel32 << 32) | kSentinel32;

// device-scope relaxed accesses: write-through stores / L1-bypassing loads (MI355X_MICROARCH.md, inter-workgroup visibility)
template <typename U>
__device__ __forceinline__ U ld_agent(const U* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename U>
__device__ __forceinline__ void st_agent(U* p, U v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// cost (and g.d) partial number idx of the evaluation
template <bool WD, typename ArgsT>
__device__ __forceinline__ void put_partial(const ArgsT& A, size_t idx, double c, double d) {
  if (A.mfinish) {
    st_agent(&A.mpart[idx], c);
    if (WD) st_agent(&A.mpart_gd[idx], d);
  } else {
    A.partials[idx] = c;
    if (WD) A.partials_gd[idx] = d;
  }
}

// In-kernel finish, executed by ONE workgroup (the last of the grid: dispatched last, among the last to finish) after
// its own work: poll until no granule holds the sentinel, add the granules in index order (deterministic), re-arm
// them, publish the cost.  Every other workgroup just publishes and leaves: no ticket, no wait on its own stores.
template <bool WD, int NT, typename ArgsT>
__device__ __forceinline__ void finish_block(const ArgsT& A, double* red /* LDS, 2 * NT / 64 doubles */) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  double tot[2] = {0.0, 0.0};
  bool timed_out = false;
  for (int pass = 0; pass < (WD ? 2 : 1); ++pass) {
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(pass == 0 ? A.mpart : A.mpart_gd);
    double acc = 0.0;
    for (int base = 0; base < A.n_partials; base += NT * 8) {
      unsigned long long a[8];
      unsigned spins = 0;
      while (true) {
        bool missing = false;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = base + u * NT + tid;
          a[u] = ld_agent(&src[i < A.n_partials ? i : base]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = base + u * NT + tid;
          missing |= (i < A.n_partials) && a[u] == kSentinel;
        }
        if (!__syncthreads_or(missing)) break;
        // bounded like the host's wait (~2 s): a workgroup that never publishes (a fault elsewhere on the device) ends the
        // evaluation with a NaN cost instead of hanging the stream; a busy or shared GPU does not get near it
        if (++spins > (1u << 21)) { timed_out = true; break; }
        if (spins < 64) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(32);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * NT + tid;
        acc += (i < A.n_partials) ? __longlong_as_double((long long)a[u]) : 0.0;
      }
    }
    if (pass == 0 && A.n_xpart > 0) {  // partials of the earlier launch: behind the granules, strided in index order
      for (int i = tid; i < A.n_xpart; i += NT) acc += A.xpart[i];
    }
    acc = wave_sum_d(acc);
    __syncthreads();
    if (lane == 0) red[wid] = acc;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    tot[pass] = t;
  }
  // re-arm: the next evaluation finds every granule unpublished.  NOT after a time-out: a workgroup that arrives late
  // would publish into a re-armed slot and the next evaluation would consume that stale partial; the granules stay as
  // they are, cost_out[0] is NaN, the sticky word cost_out[6] tells the host to re-initialise them (srmap_api.hip).
  if (timed_out) {
    if (tid == 0) { A.cost_out[0] = __builtin_nan(""); A.cost_out[6] = 1.0; if (A.to_host != nullptr) *(volatile double*)A.to_host = 1.0; if (WD && A.pub != nullptr) { A.pub[0] = A.cost_out[0]; A.pub[1] = 0.0; __threadfence_system(); *(volatile double*)A.tag_slot = A.tag; } }
    return;
  }
  for (int i = tid; i < A.n_partials; i += NT) {
    st_agent(reinterpret_cast<unsigned long long*>(A.mpart) + i, kSentinel);
    if (WD) st_agent(reinterpret_cast<unsigned long long*>(A.mpart_gd) + i, kSentinel);
  }
  if (tid == 0) {
    const double v = tot[0];
    A.cost_out[0] = v;
    if (WD) {
      A.cost_out[1] = tot[1];
      if (A.pub != nullptr) {  // solver line search: {cost, g.d} straight to the host-mapped words, then the arrival tag
        A.pub[0] = v;
        A.pub[1] = tot[1];
        __threadfence_system();
        *(volatile double*)A.tag_slot = A.tag;
      }
    }
  }
}

template <typename T> struct BorderArgs;

template <typename T, int B, int NP>
struct ZArgs {
  // Field order = order of first use: what a workgroup needs before its first request shares the first cache lines of
  // the argument block (each separately fetched line was a scalar-load round trip at the head of every workgroup).
  const T* x;
  const T* y;
  const T* w;        // IRLS weights or nullptr
  T* g;              // nullptr = cost only
  int W, H, wl, hl;
  int nby;           // grid rows (blockIdx.y) taken by border blocks; 0 = none
  int E;             // max |shift| (edge tiles take the masked code path)
  int terms;         // SRMAP_TERM_*
  int obs_C;         // channels of the observation stack (y already points at the evaluation's first channel)
  // ---- 64 bytes ----
  int cr0, cr1;      // HR rows whose cost terms are counted (row-band sharding; default 0, H)
  int rr0, rr1;      // HR rows (tile aligned) whose regulariser terms are evaluated (frame sharding; default 0, H)
  int sel_mode, sel0, sel1;  // 0: every tile; 1: only tile rows [sel0, sel1) and no border blocks (they read no halo
                             // row of x: row shards run them under the halo exchange); 2: everything else
  int n_tile_partials;  // border partials are stored behind the tile partials
  int MS;                // slots per (row phase, column phase); tables are [MS][S][S] (round, row phase, column phase)
  // the frame table by value (kernel-argument segment: always scalar loads, no table round trip before the first request):
  int cntk[4][8];        //   cnt; [pr][S + 1] = min over the column phases (rounds below it need no per-pixel test)
  long long off0[4][4];  //   round 0 of off
  T blur3[3];        // k * k^T (blur_module.cpp:20-22) of the symmetric kernel: corner, edge, centre (B == 1: 1, 1, 1)
  T k1s[2];          // the separable factor (B^T z is evaluated as two 1-D passes): outer tap, centre tap
  T lambda;
  T powtab[NP];      // BTV alpha^(i+j)
  T pwsum;           // BTV: sum of alpha^(i+j) over the gradient's (exclusive) window
  double* partials;
  const T* dvec;         // WD instances: search direction d; the kernel also produces partials of g.d
  double* partials_gd;   //   (same indexing as partials)
  const BorderArgs<T>* bd;  // device-resident constants of the border blocks
  // ---- edge tiles, later rounds, sub-pixel instances ----
  ZEntry aux0[4][4];     // round 0 of aux (edge tiles)
  const int* cnt;        // [S][8]  residuals per (row phase, column phase); [pr][S] = max over the column phases
  const long long* off;  // [MS][S][S] element offset of the observation relative to (channel plane + LR cell row * w + cell)
  const ZEntry* aux;     // [MS][S][S] the same residuals as (frame, LR row offset, LR column offset) for edge tiles
  const T* rbuf;         // SP instances (sub-pixel shifts): residuals r_k = A_k x - y_k, [K][C][h][w], from k_forward_direct
  const double* spw;     //   bilinear tap weight of every table entry, [MS][S][S]
  const ZSrc* spsrc;     //   the same taps source-major: [S][spmax] records, spn[pr] of them in use (z_row_sp2)
  int spn[4];
  int spmax;
  int Dr;                //   data gradient of the pixels within Dr of the image edge comes from the exact ring pass
  const T* ringbuf;      //   ... which ran AHEAD of this launch into ringbuf[C][2 Dr W + 2 Dr (H - 2 Dr)] (k_gather_ring's pixel
                         //   order): added as g is stored.  nullptr: the ring pass follows the launch and adds to g itself
  RingRects ring;        // border frame rectangles (border blocks / tasks, corrections)
  // ---- in-kernel finish (no second launch): partials leave as write-through granules, the last block of the grid
  // gathers them (see m_finish_block) ----
  int mfinish;           // 1: granules + in-kernel reduction; 0: plain partials, reduced by k_finish_eval / the caller
  int n_partials;        // granules of the evaluation
  double* mpart;         // cost granules [n_partials]; the sentinel pattern = not yet published
  double* mpart_gd;      // g.d granules [n_partials] (WD)
  double* cost_out;      // [0] cost, [1] g.d
  double* pub;           // solver line search: host-mapped {cost, g.d}, then the arrival tag
  double* tag_slot;
  double tag;
  double* to_host;       // host-mapped word raised when the in-kernel finish times out (solver: the solve ends at once)
  const double* xpart;   // plain cost partials of an EARLIER launch on the stream (sub-pixel path: the forward kernel's data
  int n_xpart;           //   cost), complete when this kernel starts: the in-kernel finish adds them, in index order
  // ---- solver line search (WD instances): the trial point x = fold_xk + fold_stp * dvec is formed when the window goes
  // to LDS (the expression of solver.hip's k_axpy_out, same contraction), its own pixels are written to fold_x ----
  const T* fold_xk;      // nullptr: x is read as given
  T* fold_x;
  T fold_stp;
  const double* fold_norms;  // device {max|dk|, dk.dk}: dvec is the UNNORMALISED direction, every element read from it is
                             // scaled to the normalised one (cg_norm.hpp) -- the solver stores no normalised vector
};

// The x tile is staged PRE-SCALED by 2^Q (exact: a power of two).  Every difference of two staged values is the
// true difference times 2^Q, large enough that sgn(d) * pw is just a clamp to [-pw, pw] (no ldexp per tap), and
// the scale drops out of the sums exactly (r = r' * 2^-Q, B x = (B x') * 2^-Q).  Exact for |d| >= 2^-Q and pixel
// magnitudes below 2^(Emax - Q - 3).  Q sits in the MIDDLE of the exponent range -- 512 (f64), 64 (f32) -- so the
// input domain is |x| < 2^508 (f64; the reference's own cost, a sum of r^2, overflows beyond 2^511) / 2^60 (f32), and
// differences down to 2^-512 / 2^-64 are still told from 0 (include/srmap.h, "Input domain").
template <typename T> struct Pre;
template <> struct Pre<double> {
  static constexpr int Q = 512;
  static __device__ __forceinline__ double up(double v) { return __builtin_ldexp(v, Q); }
  static __device__ __forceinline__ double down(double v) { return __builtin_ldexp(v, -Q); }
};
template <> struct Pre<float> {
  static constexpr int Q = 64;
  static __device__ __forceinline__ float up(float v) { return __builtin_ldexpf(v, Q); }
  static __device__ __forceinline__ float down(float v) { return __builtin_ldexpf(v, -Q); }
};
// sgn(d) * pw for a pre-scaled difference dq = d * 2^Q: clamp (pw <= 1 << 2^Q * |d| for every d the reference
// distinguishes from 0).
template <typename T>
__device__ __forceinline__ T sgn_pre(T dq, T pw) { return __builtin_fmin(__builtin_fmax(dq, -pw), pw); }
template <>
__device__ __forceinline__ float sgn_pre<float>(float dq, float pw) { return __builtin_amdgcn_fmed3f(dq, -pw, pw); }
// (sgn(d) + 1) / 2 of a pre-scaled difference: dq + 0.5 clamped to [0, 1] -- ONE instruction (v_add_f64 ... clamp):
// 0 / 0.5 / 1 for d < 0 / d == 0 / d > 0 (|dq| >= 1 whenever d != 0)
template <typename T>
__device__ __forceinline__ T step_pre(T dq) { return __builtin_fmin(__builtin_fmax(dq + T(0.5), T(0)), T(1)); }

// ---- index helpers: `col` is a pixel column relative to the first pixel of the thread's cell ----
template <typename C>
__device__ __forceinline__ constexpr int xi(int row, int col) {
  return row * C::XROW + posmod(col, C::TW / C::CW) * C::XC + C::XCL + floordiv(col, C::TW / C::CW);
}
template <typename C>
__device__ __forceinline__ constexpr int ci(int row, int col) {
  return row * C::CROW + posmod(col, C::TW / C::CW) * C::CC + C::CCL + floordiv(col, C::TW / C::CW);
}

// Read-only tables (frame table rounds > 0, edge-tile entries, sub-pixel tap weights) are read through the CONSTANT
// address space: a uniform load from it is a scalar load wherever it stands -- from a plain global pointer the
// compiler demotes uniform loads to vector loads + v_readfirstlane once the kernel has stored anything.
template <typename U>
__device__ __forceinline__ U ctab(const U* p, size_t i) {
  typedef const U __attribute__((address_space(4))) * CP;
  return ((CP)(unsigned long long)p)[i];
}
__device__ __forceinline__ ZEntry ctab(const ZEntry* p, size_t i) {
  typedef const int __attribute__((address_space(4))) * CP;
  CP q = (CP)(unsigned long long)(p + i);
  ZEntry e;
  e.k = q[0]; e.io = q[1]; e.jo = q[2]; e.oyx = q[3];
  return e;
}

// blur tap (a, e) of the symmetric B x B kernel from its three distinct values
template <int B, typename ArgsT>
__device__ __forceinline__ auto blur_tap(const ArgsT& A, int a, int e) {
  if (B == 1) return A.blur3[2];
  const bool ca = a == (B - 1) / 2, ce = e == (B - 1) / 2;
  return (ca && ce) ? A.blur3[2] : ((ca || ce) ? A.blur3[1] : A.blur3[0]);
}
template <int B, typename ArgsT>
__device__ __forceinline__ auto k1_tap(const ArgsT& A, int a) { return (B == 1 || a == (B - 1) / 2) ? A.k1s[1] : A.k1s[0]; }

// Round-0 entry of the edge-tile table, field by field (the argument block may live in the constant address space, from
// which a struct cannot be copied as a whole).
template <typename ArgsT>
__device__ __forceinline__ ZEntry aux0_at(const ArgsT& A, int pr, int pc) {
  ZEntry e;
  e.k = A.aux0[pr][pc].k; e.io = A.aux0[pr][pc].io; e.jo = A.aux0[pr][pc].jo; e.oyx = A.aux0[pr][pc].oyx;
  return e;
}

// Observation of LR pixel (i, j) of frame plane yk, address clamped into the image.
template <typename T>
__device__ __forceinline__ T obs_at(const T* __restrict__ yk, int i, int j, int hl, int wl) {
  const int ic = i < 0 ? 0 : (i >= hl ? hl - 1 : i);
  const int jc = j < 0 ? 0 : (j >= wl ? wl - 1 : j);
  return yk[(size_t)ic * wl + jc];
}

// Row phase / LR cell row of HR row gr (floor division; gr may be negative in the top halo).
template <int S>
__device__ __forceinline__ void row_phase(int gr, int& rc, int& pr) {
  rc = (gr >= 0) ? gr / S : -((-gr + S - 1) / S);
  pr = gr - rc * S;
}

// Observations of the t-th residual of each of the NV pixels of the thread's cell in an HR row of phase pr
// (frame table: SURVEY.md section 8a' restated per HR pixel).  Interior tiles: one scalar offset per pixel phase,
// address = row base + offset + cell.  EDGE: explicit (frame, LR row, LR column), clamped into the image.
template <typename T, int S, typename C, bool EDGE, typename ArgsT>
__device__ __forceinline__ void load_obs_row(const ArgsT& A, int pr, int rc, int t, int cell0, int lane,
                                             const T* __restrict__ ybase, const int (&cn)[S], T (&yv)[C::NV]) {
  constexpr int HB = C::HB, NV = C::NV;
  const size_t slot = (size_t)(t * S + pr) * S;  // uniform; round-major: round 0 (the prefetch) needs no table size
  const T* yrow = ybase + ((long long)rc * A.wl + cell0);  // uniform: LR cell row rc, first cell of the tile
  if (!EDGE) {
    // branch-free: the S offsets of this table row come as ONE scalar load and every pixel's observation is
    // requested (unused slots hold offset 0 = a valid element of this LR row; their values are never consumed).
    // A uniform branch per pixel made each request wait for its own scalar load: six serialised round trips per row.
    long long offs[S];
#pragma unroll
    for (int pc = 0; pc < S; ++pc) offs[pc] = (t == 0) ? A.off0[pr][pc] : ctab(A.off, slot + pc);  // t: uniform
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int pcv = v - HB, pc = posmod(pcv, S), dc = floordiv(pcv, S);
      const T* yp = yrow + (offs[pc] + dc);  // uniform pointer; the lane adds its (non-negative) cell index
#if defined(SRMAP_EXP_NTLOAD) && SRMAP_EXP_NTLOAD
      yv[v] = __builtin_nontemporal_load(&yp[(unsigned)lane]);
#else
      yv[v] = yp[(unsigned)lane];
#endif
    }
    return;
  }
  // EDGE: the same without branches -- (frame, LR row, LR column) entries of the table row in one scalar load, every
  // address clamped into the image (unused slots are frame 0, offset 0); validity is the consumer's business
  ZEntry ent[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) ent[pc] = (t == 0) ? aux0_at(A, pr, pc) : ctab(A.aux, slot + pc);  // t: uniform
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int pcv = v - HB, pc = posmod(pcv, S), dc = floordiv(pcv, S);
    const ZEntry e = ent[pc];
    yv[v] = obs_at<T>(ybase + (size_t)e.k * A.obs_C * ((size_t)A.wl * A.hl), rc + e.io, cell0 + lane + dc + e.jo, A.hl, A.wl);
  }
  (void)cn;
}

// ---- data term, phase 1, for the S pixels of one cell in tile row `rowrel` (wave-uniform) ----
// B x at NV pixels, residuals of the frames whose LR grid hits each pixel, z; returns z (B == 1) or writes the
// horizontal half of B^T z to LDS (B == 3).  `count`: the row is owned by this tile (cost is counted).
// EDGE: tiles near the image border (and partial tiles) -- LR validity masks, in-image masks and the dropped blur
// taps of LR row 0 / column 0.  The staged x is pre-scaled by 2^Q: residual = (B x') * 2^-Q - y.
template <typename T, int S, int B, typename C, bool EDGE, typename ArgsT>
__device__ __forceinline__ void z_row(const ArgsT& A, const T* __restrict__ xs, T* __restrict__ zs, int rowrel,
                                      int R0, int cell0, int lane, const T* __restrict__ ybase, bool use_pre,
                                      const T (&ypre)[C::NV], bool count, const T (&mk)[S], T (&zout)[S],
                                      double& cost) {
  constexpr int HB = C::HB, NV = C::NV;
  int rc, pr;
  row_phase<S>(R0 + rowrel, rc, pr);
  const int xrow = rowrel + C::HU;
  T bx[NV], btop[NV], bleft[NV], bcorner[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) { bx[v] = T(0); btop[v] = T(0); bleft[v] = T(0); bcorner[v] = T(0); }
#pragma unroll
  for (int a = 0; a < B; ++a) {
    T xr[NV + B - 1];
#pragma unroll
    for (int j = 0; j < NV + B - 1; ++j) xr[j] = xs[xi<C>(xrow + a - HB, j - 2 * HB) + lane];
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
    pin<4>(bx);
  }
  const T unscale = Pre<T>::down(T(1));
  int cn[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) cn[pc] = A.cntk[pr][pc];
  const int mmax = A.cntk[pr][S];
  const int mfull = EDGE ? 0 : A.cntk[pr][S + 1];  // rounds in which every column phase owns a residual
  T z[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) z[v] = T(0);
  for (int t = 0; t < mmax; ++t) {
    T yv[NV];
    if (t == 0 && use_pre) {
#pragma unroll
      for (int v = 0; v < NV; ++v) yv[v] = ypre[v];
    } else {
      load_obs_row<T, S, C, EDGE>(A, pr, rc, t, cell0, lane, ybase, cn, yv);
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
          // no such LR pixel (row: uniform, column: per lane)
          rr = ((unsigned)i < (unsigned)A.hl && (unsigned)j < (unsigned)A.wl) ? rr : T(0);
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
      zs[(rowrel + HB) * C::ZROW + pc * C::CW + lane] = zh;
    }
  }
}

// ---- data term, phase 1, SUB-PIXEL shifts ----
// The transpose warp of a sub-pixel shift is a 4-tap bilinear gather (motion_module.cpp:40-51: warpAffine with the
// negated shift); in the interior it commutes with B^T like the integer shift does, so
//     z(p) = sum_k sum_b w'_{k,b} [ (p + o'_k + tap_b) on the LR grid ] r_k((p + o'_k + tap_b) / S),   g_data = 2 S^2 B^T z.
// The residuals come from k_forward_direct (exact 4-tap forward warp, every clip); which (frame, tap) pairs hit a
// pixel depends only on its phase: host-built table (frame, LR row / column offset, weight).  No cost here (the
// forward kernel counts it); the pixels within Dr of the edge are evaluated by the exact ring pass instead.
// EDGE = false: tiles whose table rows all stay inside the LR image -- no uniform branches, one table row per round
// (padding entries carry weight 0 and a harmless in-range offset).  Columns: the address is clamped and the WEIGHT
// masked per lane (nothing is done to the loaded value before the multiply-add, so a round's loads stay in flight
// together).
// COLCLAMP = false (with EDGE = false): tile columns whose table columns all stay inside the LR image -- the address
// is a uniform base + the lane, the weight a scalar operand of the multiply-add (no per-lane index arithmetic at all).
template <typename T, int S, typename C, bool EDGE, bool COLCLAMP, typename ArgsT>
__device__ __forceinline__ void sp_load_round(const ArgsT& A, int pr, int rc, int t, int cell0, int lane, int ch,
                                              const int (&cn)[S], T (&rv)[C::NV], T (&wm)[C::NV]) {
  constexpr int HB = C::HB, NV = C::NV;
  const size_t nl = (size_t)A.wl * A.hl;
  const int slot = (t * S + pr) * S;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int pcv = v - HB, pc = posmod(pcv, S), dc = floordiv(pcv, S);
    rv[v] = T(0);
    wm[v] = T(0);
    if (!EDGE || t < cn[pc]) {  // uniform
      const ZEntry e = ctab(A.aux, (size_t)(slot + pc));
      const int i = rc + e.io, j = cell0 + lane + dc + e.jo;
      if (!EDGE || (unsigned)i < (unsigned)A.hl) {  // uniform
        const T* plane = A.rbuf + (size_t)(e.k * A.obs_C + ch) * nl;
        if (!EDGE && !COLCLAMP) {
          const T* rowp = plane + ((long long)i * A.wl + (cell0 + dc + e.jo));  // uniform
          rv[v] = rowp[(unsigned)lane];
          wm[v] = (T)ctab(A.spw, (size_t)(slot + pc));
        } else {
          const int jc = j < 0 ? 0 : (j >= A.wl ? A.wl - 1 : j);
          rv[v] = plane[i * A.wl + jc];
          wm[v] = ((unsigned)j < (unsigned)A.wl) ? (T)ctab(A.spw, (size_t)(slot + pc)) : T(0);
        }
      }
    }
  }
}

template <typename T, int S, int B, typename C, bool EDGE, bool COLCLAMP, typename ArgsT>
__device__ __forceinline__ void z_row_sp(const ArgsT& A, T* __restrict__ zs, int rowrel, int R0, int cell0, int lane, int ch,
                                         T (&zout)[S]) {
  constexpr int HB = C::HB, NV = C::NV;
  int rc, pr;
  row_phase<S>(R0 + rowrel, rc, pr);
  int cn[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) cn[pc] = A.cntk[pr][pc];
  const int mmax = A.cntk[pr][S];
  T z[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) z[v] = T(0);
  for (int t = 0; t < mmax; ++t) {
    T rv[NV], wm[NV];
    sp_load_round<T, S, C, EDGE, COLCLAMP>(A, pr, rc, t, cell0, lane, ch, cn, rv, wm);
#pragma unroll
    for (int v = 0; v < NV; ++v) z[v] += wm[v] * rv[v];
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
      zs[(rowrel + HB) * C::ZROW + pc * C::CW + lane] = zh;
    }
  }
}

// The same sum with the taps grouped by SOURCE (frame, vertical tap): the residual r_k(i, m) a horizontal tap pair lands
// on serves two neighbouring pixels -- (m S - ox) with the dx = 0 weight and the pixel left of it with the dx = 1 weight --
// so a cell's S + 2 HB pixels need 1 - 2 residuals per source instead of one request per (pixel, tap): 14 requests per
// row instead of 24 - 30 at cfg2's 16 frames (entry-major rounds are padded to the longest phase).  With
// a = (-ox) mod S the pixels are pcv = a + S e (dx = 0) and a - 1 + S e (dx = 1), the residual column (ox + a) / S + e,
// e in {-1, 0, 1}: compile-time pixel indices per value of a (uniform switch).
// Requests and arithmetic are separated per chunk of kSpChunk sources: the chunk's table records are loaded back to back,
// then all its residual requests are issued (uniform frame base + a 32-bit row / column / lane offset), then the
// multiply-adds run per source with compile-time pixel indices (uniform switch on a).  The scalar unit of a CU is shared
// by its sixteen waves: the phase clock showed this gather at 8 - 12 K cycles per row while it held one table round trip,
// a 64-bit address chain and three column tests per source there (profiles/r05_subpixel.txt).
constexpr int kSpChunk = 4;
template <typename T, int S, typename C, int A, bool COLCLAMP, typename ArgsT>
__device__ __forceinline__ void sp_apply(const ArgsT& A_, const T (&rv)[kSpSlots(S, C::HB)], int elo, int jbase, int lane,
                                         T w0, T w1, T (&z)[C::NV]) {
  constexpr int HB = C::HB, NSL = kSpSlots(S, C::HB);
  // e_lo as the host derived it for this a (compile-time here): the first column any pixel of the cell uses
  constexpr int ELO = ((A - S >= -HB) || (A - 1 - S >= -HB)) ? -1 : 0;
  (void)elo;
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    const int e = ELO + sl;
    if (e > 1) continue;
    const int p0 = A + S * e, p1 = A - 1 + S * e;                 // pixel (relative to the cell) of the dx = 0 / dx = 1 tap
    const bool in0 = p0 >= -HB && p0 < S + HB, in1 = p1 >= -HB && p1 < S + HB;
    if (!in0 && !in1) continue;
    T m0 = w0, m1 = w1;
    if (COLCLAMP) {  // the WEIGHT is masked per lane: nothing was done to the loaded value
      const bool ok = (unsigned)(jbase + lane + e) < (unsigned)A_.wl;
      m0 = ok ? w0 : T(0);
      m1 = ok ? w1 : T(0);
    }
    if (in0) z[(in0 ? p0 : 0) + HB] += m0 * rv[sl];
    if (in1) z[(in1 ? p1 : 0) + HB] += m1 * rv[sl];
  }
}

template <typename T, int S, int B, typename C, bool EDGE, bool COLCLAMP, typename ArgsT>
__device__ __forceinline__ void z_row_sp2(const ArgsT& A, T* __restrict__ zs, int rowrel, int R0, int cell0, int lane, int ch,
                                          T (&zout)[S]) {
  constexpr int HB = C::HB, NV = C::NV, NSL = kSpSlots(S, C::HB);
  int rc, pr;
  row_phase<S>(R0 + rowrel, rc, pr);
  const size_t nl = (size_t)A.wl * A.hl;
  const int ns = A.spn[pr];   // a multiple of kSpChunk (null records behind the last source)
  T z[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) z[v] = T(0);
  typedef const ZSrc __attribute__((address_space(4))) * SrcPtr;
  SrcPtr tab = (SrcPtr)(unsigned long long)(A.spsrc + (size_t)pr * A.spmax);
  for (int n0 = 0; n0 < ns; n0 += kSpChunk) {
    // ---- the chunk's records (scalar loads, no control flow in between) ----
    int rk[kSpChunk], rio[kSpChunk], rq[kSpChunk], ram[kSpChunk];
    double rw0[kSpChunk], rw1[kSpChunk];
#pragma unroll
    for (int c = 0; c < kSpChunk; ++c) {
      rk[c] = tab[n0 + c].k; rio[c] = tab[n0 + c].io; rq[c] = tab[n0 + c].q; ram[c] = tab[n0 + c].am;
      rw0[c] = tab[n0 + c].w0; rw1[c] = tab[n0 + c].w1;
    }
    // ---- requests of the chunk ----
    T rv[kSpChunk][NSL];
    int as[kSpChunk], jb[kSpChunk];
#pragma unroll
    for (int c = 0; c < kSpChunk; ++c) {
      const int i = rc + rio[c];
      as[c] = ram[c] & 0xff;
      if (EDGE && (unsigned)i >= (unsigned)A.hl) as[c] = kSpNull;   // uniform: no such LR row
      jb[c] = cell0 + rq[c];
      const int elo = (ram[c] >> 8) - 1;
      // null records (and rows outside the image) request the frame's element 0: harmless, never used
      const bool live = as[c] != kSpNull;
      const T* base = A.rbuf + (size_t)((live ? rk[c] : 0) * A.obs_C + ch) * nl;   // uniform
      const unsigned rowoff = live ? (unsigned)i * (unsigned)A.wl : 0u;
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl) {
#ifdef SRMAP_EXP_SPNOLOAD
        rv[c][sl] = (T)(lane + sl) * (T)rw0[c];                 // TIMING ONLY: no residual request
#else
        if (!COLCLAMP) {
          rv[c][sl] = base[rowoff + (unsigned)(live ? jb[c] + elo + sl + lane : 0)];
        } else {
          const int j = jb[c] + lane + elo + sl;
          rv[c][sl] = base[rowoff + (unsigned)(live ? (j < 0 ? 0 : (j >= A.wl ? A.wl - 1 : j)) : 0)];
        }
#endif
      }
    }
    // ---- multiply-adds, per source with compile-time pixel indices ----
#pragma unroll
    for (int c = 0; c < kSpChunk; ++c) {
      const int a = as[c];
      const int elo = (ram[c] >> 8) - 1;
      const T w0 = (T)rw0[c], w1 = (T)rw1[c];
      if (a == 0) sp_apply<T, S, C, 0, COLCLAMP>(A, rv[c], elo, jb[c], lane, w0, w1, z);
      else if (a == 1) sp_apply<T, S, C, 1, COLCLAMP>(A, rv[c], elo, jb[c], lane, w0, w1, z);
      else if (a == 2) sp_apply<T, S, C, (S > 2 ? 2 : 0), COLCLAMP>(A, rv[c], elo, jb[c], lane, w0, w1, z);
      else if (a == 3) sp_apply<T, S, C, (S > 3 ? 3 : 0), COLCLAMP>(A, rv[c], elo, jb[c], lane, w0, w1, z);
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
      zs[(rowrel + HB) * C::ZROW + pc * C::CW + lane] = zh;
    }
  }
}

// t = 0 observations of the NV pixels of the thread's cell in tile row `rowrel`, issued at kernel start.
template <typename T, int S, int B, typename C, typename ArgsT>
__device__ __forceinline__ void z_row_prefetch(const ArgsT& A, int rowrel, int R0, int cell0, int lane, bool edge,
                                               const T* __restrict__ ybase, T (&ypre)[C::NV]) {
  int rc, pr;
  row_phase<S>(R0 + rowrel, rc, pr);
  int cn[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) cn[pc] = A.cntk[pr][pc];
  if (edge) load_obs_row<T, S, C, true>(A, pr, rc, 0, cell0, lane, ybase, cn, ypre);
  else load_obs_row<T, S, C, false>(A, pr, rc, 0, cell0, lane, ybase, cn, ypre);
}

// ---- regulariser pass 1 for the S pixels of one cell (tv_regularizer.cpp:110-170, btv_regularizer.cpp:19-136) ----
// FULL: values, self term into acc, cost, 2*lambda*w*r into cs (own rows).  !FULL: 2*lambda*w*r only (halo rows).
// Stores 0 for pixels outside the image and, BTV only, for the absolute pixel (0,0) (btv_regularizer.cpp:143-146).
template <typename T, int S, int REGK, int R, typename C, bool BORDER, bool FULL>
__device__ __forceinline__ void reg_row(T (&acc)[S], double& cost, const T* __restrict__ xs, T* __restrict__ cs,
                                        const T (&wv)[S], int rowrel, int lane, int gr, int gc0, int W, int H,
                                        T lambda, const T (&pw)[C::NP], T pwsum, bool cost_row) {
  constexpr int WIN = C::WIN;
  constexpr int NC = S + WIN;
  const int xrow = rowrel + C::HU;
  T x0v[S], rv[S], dv[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) { rv[pc] = T(0); dv[pc] = T(0); }
  // BORDER (tiles at the right / bottom image edge): a tap beyond the image is the reference's skipped tap = a zero
  // difference.  Rows: gr + i is wave-uniform -- a window row below the image is handled as a whole.  Columns: the image
  // width and a thread's first column are multiples of S, so a tap column c = pc + j of an in-image cell lies beyond the
  // image only if it crosses into a FOLLOWING cell (c >= S; with S = 2 the window reaches two cells on) and that cell does:
  // a per-lane factor (1 / 0) per following cell on the crossing taps instead of two compares and two selects on every tap
  // (the masked path cost the edge tiles 148 compare / select issues per wave; d * 1 = d and |d * 0| = 0,
  // step(+-0) = 1/2: the same bits as the select).
  static_assert(S - 1 + WIN < 3 * S, "a window tap reaches at most two cells beyond the thread's own");
  const T mR1 = (!BORDER || gc0 + S < W) ? T(1) : T(0), mR2 = (!BORDER || gc0 + 2 * S < W) ? T(1) : T(0);
  // the window is walked row by row (i outer, j inner per pixel: the reference's summation order)
#pragma unroll
  for (int i = 0; i <= WIN; ++i) {
    if (BORDER && REGK == 2 && i > 0 && gr + i >= H) {  // uniform: every tap of this window row is skipped
      if (FULL && sizeof(T) == 8 && i < R) {
#pragma unroll
        for (int pc = 0; pc < S; ++pc) {
#pragma unroll
          for (int j = 0; j < R; ++j) dv[pc] += pw[i + j] * T(0.5);   // (sgn(0) + 1) / 2, as the masked path adds it
        }
      }
      continue;
    }
    T row[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) row[j] = xs[xi<C>(xrow + i, j) + lane];
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
          if (BORDER && pc + j >= S) d = d * ((pc + j) / S == 1 ? mR1 : mR2);  // skipped tap == zero difference
          rv[pc] += pw[i + j] * absv(d);
          if (FULL && i < R && j < R) {  // exclusive window in the gradient
            if (sizeof(T) == 8) dv[pc] += pw[i + j] * step_pre<T>(d);  // (sgn + 1) / 2: add with clamp + FMA, two f64 issues
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
    pin<1>(rv);
    if (FULL) pin<1>(dv);
  }
#pragma unroll
  for (int pc = 0; pc < S; ++pc) {
    const T r = Pre<T>::down(rv[pc]);  // the staged x is pre-scaled: r = r' * 2^-Q exactly
    const T c = lambda * wv[pc];
    T cr2 = T(2) * c * r;
    const bool in_img = (unsigned)gr < (unsigned)H && (unsigned)(gc0 + pc) < (unsigned)W;
    if (FULL) {
      if (REGK == 2 && sizeof(T) == 8) dv[pc] = T(2) * dv[pc] - pwsum;  // sum pw * sgn = 2 * sum pw * (sgn + 1) / 2 - sum pw
      acc[pc] += cr2 * dv[pc];
      const double cd = (in_img && cost_row) ? (double)c * (double)r * (double)r : 0.0;
      cost += cd;
    }
    if (!in_img || (REGK == 2 && gr == 0 && gc0 + pc == 0)) cr2 = T(0);
    cs[ci<C>(rowrel + C::RU, pc) + lane] = cr2;
  }
}

// 2*lambda*w*r of ONE pixel at tile-relative (rowrel (per lane), col (compile time, < 0)): the left halo columns.
// xs / cs arrive already offset by the lane's row (rowrel * XROW / rowrel * CROW): every index below is an immediate.
// BORDER = false: the window stays inside the image (no per-tap masks).
template <typename T, int S, int REGK, int R, typename C, int COL, bool BORDER>
__device__ __forceinline__ void reg_halo_col(const T* __restrict__ xs, T* __restrict__ cs, const T wt,
                                             int rowrel, int R0, int C0, int W, int H, T lambda,
                                             const T (&pw)[C::NP]) {
  constexpr int WIN = C::WIN;
  const int gr = R0 + rowrel, gc = C0 + COL;
  constexpr int xrow = C::HU;
  T cr2 = T(0);
  if (gr >= 0 && gr < H && gc >= 0 && gc < W && !(REGK == 2 && gr == 0 && gc == 0)) {
    const T x0 = xs[xi<C>(xrow, COL)];
    T r = T(0);
    if (REGK == 2) {
#pragma unroll
      for (int i = 0; i <= WIN; ++i) {
#pragma unroll
        for (int j = 0; j <= WIN; ++j) {
          if (i == 0 && j == 0) continue;
          const T v = xs[xi<C>(xrow + i, COL + j)];
          const T d = (!BORDER || (gr + i < H && gc + j < W)) ? x0 - v : T(0);
          r += pw[i + j] * absv(d);
        }
      }
    } else {
      const T yv = (!BORDER || gr + 1 < H) ? absv(xs[xi<C>(xrow + 1, COL)] - x0) : T(0);
      const T xv = (!BORDER || gc + 1 < W) ? absv(xs[xi<C>(xrow, COL + 1)] - x0) : T(0);
      r = yv + xv;
    }
    cr2 = T(2) * (lambda * wt) * Pre<T>::down(r);
  }
  cs[ci<C>(C::RU, COL)] = cr2;
}

// ---- regulariser pass 2: contributions of the up / left neighbours (tv_regularizer.cpp:172-203,
// btv_regularizer.cpp:137-162) ----
template <typename T, int S, int REGK, int R, typename C>
__device__ __forceinline__ void reg_pass2z(T (&acc)[S], const T* __restrict__ xs, const T* __restrict__ cs, int rowrel,
                                           int lane, const T (&pw)[C::NP]) {
  constexpr int RU = C::RU;
  if (RU == 0) return;
  constexpr int NC = S + RU;
  const int xrow = rowrel + C::HU, crow = rowrel + RU;
  T x0v[S], sum[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) sum[pc] = T(0);
#pragma unroll
  for (int i = 0; i <= RU; ++i) {  // neighbour row r - i
    T xw[NC], cw[NC];              // columns -RU .. S-1
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      xw[j] = xs[xi<C>(xrow - i, j - RU) + lane];
      cw[j] = cs[ci<C>(crow - i, j - RU) + lane];
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
    pin<2>(sum);
  }
#pragma unroll
  for (int pc = 0; pc < S; ++pc) acc[pc] += sum[pc];
}

// ---------------------------------------------------------------------------------------------------------
// Border blocks: what the frame-summed tile path cannot express, with the reference's literal per-frame formulas
// (SURVEY.md section 8a').  They are extra workgroups at the FRONT of k_eval_z's grid (dispatched first, they run
// beside the first tiles and cost no launch of their own).  One thread per pixel q of the frame of width 2E around
// the image edge, [-E, H+E) x [-E, W+E) minus [E, H-E) x [E, W-E), E = max |shift|:
//   q OUTSIDE the image: cost of the residuals whose z position is q (they have no owner thread among the tiles);
//   q INSIDE: the transpose warp clips its source (motion_module.cpp:40-51 on an H x W image): frame k reaches
//       q only if q - o_k is inside the image.  The tiles add every frame; the excluded ones are collected here,
//       corr[q] = 2 S^2 sum_tap B^T[tap] sum_{k in L(q + tap), q - o_k outside} r_k, and subtracted from g by
//       k_finish_eval after the tile kernel.
constexpr int kBorderTabEntries = 256;  // frame-table entries staged in LDS by the border blocks

template <typename T>
struct BorderArgs {      // device-resident (one per problem): only the border blocks read it
  const int2* hdr;       // flat frame table: (count, first entry) per (row phase, column phase)
  const ZEntry* ent;
  const T* blur_d;       // [b*b] device copies (dynamic indexing)
  T* corr;               // [C][n_ring]
  int S, b, hb;
  int n_ring;            // pixels of the frame
  int n_ent;             // entries of the frame table
  int obs_C;             // channels of the observation stack
};

__device__ __forceinline__ int dfdiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// r_k(i, j) = (D B M_k x)(i, j) - y_k(i, j) with both clips (warped image, blur zero padding)
// S, B at compile time and the taps from the kernel arguments: the B * B loads of a residual are requested together
// (with run-time loop bounds and a tap table in memory every tap was its own round trip: ~9 us per border block).
template <typename T, int S, int B, bool FOLD = false, typename ArgsT>
__device__ __forceinline__ T border_residual(const ArgsT& A, int W, int H, int wl, const T* __restrict__ xplane,
                                             const T* __restrict__ yk, int ox, int oy, int i, int j,
                                             const T* __restrict__ dplane = nullptr, T stp = T(0),
                                             const DirScale& ds = DirScale{0.0, 1.0, 1.0, false}) {
  constexpr int hb = (B - 1) / 2;
  T xv[B * B];
  const T yv = yk[(size_t)i * wl + j];
  // 32-bit element offsets inside the plane (the plan admits planes below 2^31 elements), column terms hoisted out of
  // the row loop, every request at a valid address (masked afterwards)
  int cix[B];
  bool cok[B];
#pragma unroll
  for (int e = 0; e < B; ++e) {
    const int cc = S * j + e - hb;
    const int sc = cc + ox;
    cok[e] = cc >= 0 && cc < W && sc >= 0 && sc < W;
    cix[e] = cok[e] ? sc : 0;
  }
#pragma unroll
  for (int a = 0; a < B; ++a) {
    const int rr = S * i + a - hb;
    const int sr = rr + oy;
    // filter2D BORDER_CONSTANT on the warped image, warpAffine BORDER_CONSTANT on the source
    const bool rok = rr >= 0 && rr < H && sr >= 0 && sr < H;
    const int rix = rok ? sr * W : 0;
#pragma unroll
    for (int e = 0; e < B; ++e) {
      // mask as a multiply: a select on the loaded value lets the compiler sink each load under its own branch
      T xval = xplane[(unsigned)(rix + cix[e])];
      if (FOLD) xval = xval + stp * dir_elem<T>(dplane[(unsigned)(rix + cix[e])], ds);   // the line search's trial point, k_axpy_out's expression
      xv[a * B + e] = xval * ((rok && cok[e]) ? T(1) : T(0));
    }
  }
  T acc = T(0);
#pragma unroll
  for (int a = 0; a < B; ++a)
#pragma unroll
    for (int e = 0; e < B; ++e) acc += blur_tap<B>(A, a, e) * xv[a * B + e];
  return acc - yv;
}

// One border block of NT threads; smem: scratch of at least 16 int2 + kBorderTabEntries ZEntry + 8 doubles.
// nbb: border blocks per channel (the block's partial is stored at n_tile_partials + ch * nbb + bidx).
template <typename T, int S, int B, int NT, bool WD, bool FOLD = false, typename ArgsT>
__device__ __forceinline__ void border_block(const ArgsT& A, const BorderArgs<T>& Bd, int bidx, int ch, void* smem, int nbb) {
  const int obs_C = Bd.obs_C;
  int2* s_hdr = reinterpret_cast<int2*>(smem);
  ZEntry* s_ent = reinterpret_cast<ZEntry*>(s_hdr + 16);
  double* red = reinterpret_cast<double*>(s_ent + kBorderTabEntries);
  const int tid = threadIdx.x;
  const int t = bidx * NT + tid;
  const size_t N = (size_t)A.W * A.H, nl = (size_t)A.wl * A.hl;
  // the frame table -> LDS (one global latency for the whole block instead of one per lookup)
  if (tid < S * S) s_hdr[tid] = Bd.hdr[tid];
  for (int i = tid; i < Bd.n_ent; i += NT) s_ent[i] = Bd.ent[i];
  __syncthreads();
  double cost = 0.0, gdc = 0.0;
  if (t < Bd.n_ring) {
    int qr, qc;
    ring_pixel(t, A.W, A.H, A.ring, qr, qc);
    const T* xplane = (FOLD ? A.fold_xk : A.x) + (size_t)ch * N;
    const T* dplane = FOLD ? A.dvec + (size_t)ch * N : nullptr;
    const T fstp = FOLD ? A.fold_stp : T(0);
    const DirScale ds = dir_scale(WD ? A.fold_norms : nullptr);
    const T* ybase = A.y + (size_t)ch * nl;
    const bool inside = qr >= 0 && qr < A.H && qc >= 0 && qc < A.W;
    T corr = T(0);
    if (!inside) {
      if (A.terms & SRMAP_TERM_DATA) {
        const int rc = dfdiv(qr, S), cc = dfdiv(qc, S);
        const int2 h = s_hdr[(qr - rc * S) * S + (qc - cc * S)];
        for (int n = 0; n < h.x; ++n) {
          const ZEntry e = s_ent[h.y + n];
          const int i = rc + e.io, j = cc + e.jo;
          if (i < 0 || i >= A.hl || j < 0 || j >= A.wl) continue;
          if (S * i < A.cr0 || S * i >= A.cr1) continue;
          const int oy = e.oyx >> 16, ox = (int)(short)(e.oyx & 0xffff);
          const double r = (double)border_residual<T, S, B, FOLD>(A, A.W, A.H, A.wl, xplane,
                                                                  ybase + (size_t)e.k * obs_C * nl, ox, oy, i, j, dplane, fstp, ds);
          cost += r * r;
        }
      }
    } else if (A.g != nullptr && (A.terms & SRMAP_TERM_DATA)) {
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
                    border_residual<T, S, B, FOLD>(A, A.W, A.H, A.wl, xplane, ybase + (size_t)e.k * obs_C * nl, ox, oy, i, j, dplane, fstp, ds);
          }
        }
      }
      corr *= (T)(2 * S * S);
      // g.d of the corrected gradient: the correction's share, over the rows whose terms this problem counts
      if (WD && qr >= A.cr0 && qr < A.cr1) gdc = -(double)corr * (double)dir_elem<T>(A.dvec[(size_t)ch * N + (size_t)qr * A.W + qc], ds);
    }
    if (A.g != nullptr) Bd.corr[(size_t)ch * Bd.n_ring + t] = corr;
  }
  // block partial (s^2 * sum of squares), stored behind the tile partials
  {
    double v = cost;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = tid & 63, wid = tid >> 6;
    if (lane == 0) red[wid] = v;
    __syncthreads();
    if (tid == 0) {
      double sum = 0.0;
      for (int i = 0; i < NT / 64; ++i) sum += red[i];
      if (!WD) put_partial<false>(A, (size_t)A.n_tile_partials + (size_t)ch * nbb + bidx, (double)(S * S) * sum, 0.0);
      else red[NT / 64] = (double)(S * S) * sum;
    }
    if (WD) {
      double w2 = gdc;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) w2 += __shfl_down(w2, o, 64);
      __syncthreads();
      if (lane == 0) red[wid] = w2;
      __syncthreads();
      if (tid == 0) {
        double sum = 0.0;
        for (int i = 0; i < NT / 64; ++i) sum += red[i];
        put_partial<true>(A, (size_t)A.n_tile_partials + (size_t)ch * nbb + bidx, red[NT / 64], sum);
      }
    }
  }
}


}  // namespace

// host side: plan (per problem, owned by the problem)
struct ZPlan {
  int S = 0, B = 1;
  int regk = 0, regr = 0, reg_index = -1;
  bool subpix = false;  // sub-pixel shifts: residuals from k_forward_direct, z by 4-tap tables, exact ring by k_gather_direct
  int Dr = 0;           //   ring width
  void* d_ringbuf = nullptr;   //   [C][ring pixels] data gradient of the ring pixels (the ring pass ahead of the tile kernel)
  double* d_spw = nullptr;
  ZSrc* d_spsrc = nullptr;     //   source-major tap table [S][spmax] (z_row_sp2)
  int spn[4] = {0, 0, 0, 0};
  int spmax = 0;
  SpForwardPlan spf;    //   forward tile kernel (kernels_spfwd.hip); the direct forward kernel when it does not apply
  int E = 0;   // max |shift|
  int MS = 1;  // table slots per (row phase, column phase)
  int n_ent = 0;
  int2* d_hdr = nullptr;       // flat table of k_border: (count, first entry) per phase
  ZEntry* d_ent = nullptr;
  int h_cnt[32] = {0};         // host copies handed to the kernel by value: [4][8] counts (+ max, min over the column phases)
  long long h_off0[16] = {0};  //   [4][4] round-0 offsets
  ZEntry h_aux0[16] = {};      //   [4][4] round-0 (frame, LR row / column offset) entries
  int* d_cnt = nullptr;        // tile kernel: [S][8]
  long long* d_off = nullptr;  //              [MS][S][S]
  ZEntry* d_aux = nullptr;     //              [MS][S][S]
  int n_ring = 0;              // pixels of the border frame (0 when no shift produces border work)
  RingRects ring = {{0, 0, 0, 0, 0, 0}};
  void* d_corr = nullptr;      // [C][n_ring] border corrections of the gradient
  void* d_bd = nullptr;        // BorderArgs<T> (device)
  double* d_mpart = nullptr;   // write-through partial granules of the in-kernel cost reduction, [2][mpart_cap] (cost, g.d); sentinel = unpublished
  size_t mpart_cap = 0;
};


// in-kernel finish of this launch; publish {cost, g.d} to the solver's host words; plain partials of an earlier launch to add
struct MFin { bool on, publish; const double* xpart; int n_xpart; };

// Kernel arguments of the tile kernel (everything but the grid-dependent fields).
template <typename T, int S, int B, int REGK, int R>
static void fill_zargs(ZArgs<T, B, ZCfg<T, S, B, REGK, R>::NP>& A, srmap_problem* p, const Geometry& geo, int obs_c0,
                       unsigned terms, const T* x, T* g, const T* wts, const ZPlan& z, double* partials, const T* dvec,
                       double* partials_gd) {
  using C = ZCfg<T, S, B, REGK, R>;
  A.x = x; A.y = (const T*)p->d_obs + (size_t)obs_c0 * geo.w * geo.h; A.w = wts; A.g = g; A.partials = partials;
  A.dvec = dvec; A.partials_gd = partials_gd;
  A.cnt = z.d_cnt; A.off = z.d_off; A.aux = z.d_aux; A.MS = z.MS;
  A.spsrc = z.d_spsrc; A.spmax = z.spmax;
  for (int pr = 0; pr < 4; ++pr) A.spn[pr] = z.spn[pr];
  for (int pr = 0; pr < 4; ++pr) {
    for (int i = 0; i < 8; ++i) A.cntk[pr][i] = z.h_cnt[pr * 8 + i];
    for (int pc = 0; pc < 4; ++pc) { A.off0[pr][pc] = z.h_off0[pr * 4 + pc]; A.aux0[pr][pc] = z.h_aux0[pr * 4 + pc]; }
  }
  A.W = geo.W; A.H = geo.H; A.wl = geo.w; A.hl = geo.h;
  A.obs_C = p->geo.C;
  A.E = z.E;
  A.ring = z.ring;
  A.cr0 = geo.cr0; A.cr1 = geo.cr1;
  A.rr0 = geo.rr0; A.rr1 = geo.rr1;
  A.terms = (int)terms;
  if (B == 1) { A.blur3[0] = A.blur3[1] = A.blur3[2] = T(1); A.k1s[0] = A.k1s[1] = T(1); }
  else {
    const int hb = (B - 1) / 2;
    A.blur3[0] = (T)p->blur2d[0]; A.blur3[1] = (T)p->blur2d[hb]; A.blur3[2] = (T)p->blur2d[hb * B + hb];
    A.k1s[0] = (T)p->blur1d[0]; A.k1s[1] = (T)p->blur1d[hb];
  }
  A.lambda = T(0);
  for (int i = 0; i < C::NP; ++i) A.powtab[i] = T(1);
  if (REGK != 0) {
    const RegSpec& rs = p->reg[z.reg_index];
    A.lambda = (T)rs.lambda;
    if (REGK == 2) for (int i = 0; i < C::NP; ++i) A.powtab[i] = (T)rs.pow_table[i];
  }
  A.pwsum = T(0);
  if (REGK == 2)
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < R; ++j)
        if (i + j > 0) A.pwsum += A.powtab[i + j];
  A.fold_xk = nullptr; A.fold_x = nullptr; A.fold_stp = T(0); A.fold_norms = nullptr;
}

}  // namespace srmap
