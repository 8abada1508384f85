// kernels_zmarch.hip -- the hot path as a MARCHING kernel: one MAP gradient iteration
// (ObjectiveFunction::ComputeAllTerms, objective_function.cpp:5-20) by ONE resident workgroup of 16 waves per CU.
//
// A workgroup owns a strip of 64 LR cells x a band of HR rows and walks down the band 16 rows per step (wave = HR row,
// lane = LR cell, a thread owns the S pixels of its cell -- as in the tile kernel, kernels_ztile.hip, whose formulation
// and per-stage clipping rules this kernel shares: DESIGN.md section 3.1).  Per step:
//   head     the x rows the NEXT step adds to the window are requested straight into LDS (global_load_lds_dwordx4 into
//            the ring slots the previous step released; no registers, no ds_write); observations / IRLS weights of
//            this step's rows into registers;
//   phase 1  per wave: regulariser pass 1 of row r (values, self term, 2*lambda*w*r -> LDS ring) and the data term of
//            row r + 1 (B x, residuals, z, horizontal half of B^T -> zh ring) from the same four window rows;
//   barrier  (every request of the head has landed by now: the wait in front of it is free)
//   phase 2  vertical half of B^T, regulariser pass 2, g store;  barrier.
// Halo rows of zh / 2*lambda*w*r are evaluated once per band by a "virtual step" in front of the first one (phase 1
// of the last waves only).  No generations of workgroups, no tail, no per-tile argument fetch / address arithmetic /
// launch ramp: profiles/r05_ceiling.txt measured what those cost the tile kernel (39.7 -> 30.8 us with loads and halo
// passes removed, and the rest is its four generations of 2048 workgroups).
// No MFMA: stencil path.

#include "zmarch_dev.hpp"

// measurement / debugging switches (tools/exp_build_m.sh); the product build leaves them at 0
#ifndef SRMAP_EXP_MDBG
#define SRMAP_EXP_MDBG 0
#endif
// SRMAP_EXP_MCLOCK: TIMING ONLY -- lanes 0 / 1 of every wave overwrite their gradient pixels of a row with the s_memtime
// stamps of the step's phase boundaries (tools/march_clock.py reads them back out of g)
#ifndef SRMAP_EXP_MCLOCK
#define SRMAP_EXP_MCLOCK 0
#endif
#define MSTAMP(k) do { if (SRMAP_EXP_MCLOCK) stamp[k] = (double)(long long)(__builtin_amdgcn_s_memtime() - tbase); } while (0)

namespace srmap {

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(lds_ptr_t)p; }
// Direct-to-LDS request: every active lane's 16 bytes at base + voff land at lds + lane * 16.  Inline assembly on
// purpose: the compiler's builtin brackets the request with s_waitcnt vmcnt(0) and waits for it in front of every
// barrier; here the one wait is placed by hand in front of the step's first barrier.
__device__ __forceinline__ void dma16(unsigned voff, const void* base, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory");
}
__device__ __forceinline__ void vm_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// workgroup barrier for LDS traffic only (no release fence over global memory: the g stores stay in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The numeric constants a phase needs, fetched from the kernel-argument segment AT THE START OF THAT PHASE through a
// pointer the compiler cannot see through: read from the by-value argument they were hoisted out of the step loop and
// held in ~40 scalar registers across it -- with the loop's own state that was 70 - 90 spilled SGPRs, a chain of
// v_writelane / v_readlane at the head of every step.
template <typename T, int B, int NP>
struct MKonst {
  T blur3[3], k1s[2], lambda, powtab[NP], pwsum;
  int W, H;
};
typedef const char __attribute__((address_space(4))) * kptr_t;
template <typename U>
__device__ __forceinline__ U kload(kptr_t kp, size_t off) {
  typedef const U __attribute__((address_space(4))) * CP;
  return *(CP)(kp + off);
}
template <typename T, int B, int NP, typename ArgsT>
__device__ __forceinline__ MKonst<T, B, NP> load_konst(kptr_t kp) {
  asm volatile("" : "+s"(kp));
  MKonst<T, B, NP> K;
#pragma unroll
  for (int i = 0; i < 3; ++i) K.blur3[i] = kload<T>(kp, offsetof(ArgsT, blur3) + i * sizeof(T));
#pragma unroll
  for (int i = 0; i < 2; ++i) K.k1s[i] = kload<T>(kp, offsetof(ArgsT, k1s) + i * sizeof(T));
  K.lambda = kload<T>(kp, offsetof(ArgsT, lambda));
#pragma unroll
  for (int i = 0; i < NP; ++i) K.powtab[i] = kload<T>(kp, offsetof(ArgsT, powtab) + i * sizeof(T));
  K.pwsum = kload<T>(kp, offsetof(ArgsT, pwsum));
  K.W = kload<int>(kp, offsetof(ArgsT, W));
  K.H = kload<int>(kp, offsetof(ArgsT, H));
  return K;
}

#ifndef SRMAP_EXP_MWG8
#define SRMAP_EXP_MWG8 2   // resident 8-wave workgroups per CU (f32 instances)
#endif
template <typename T, int S, int B, int REGK, int R, int NW, bool WD>
__global__ __launch_bounds__(64 * NW, (NW == 16 ? 4 : 2 * SRMAP_EXP_MWG8)) void k_eval_m(ZArgs<T, B, ZCfg<T, S, B, REGK, R>::NP> A) {
  using C = MCfg<T, S, B, REGK, R, NW>;
  using GT = typename Gran<T, C::G>::type;
  using ArgsT = ZArgs<T, B, ZCfg<T, S, B, REGK, R>::NP>;
  using KT = MKonst<T, B, C::NP>;
  using W0T = PWin<T, S, C::XC, C::XCL, C::P1LO, C::P1HI>;
  using W2T = PWin<T, S, C::XC, C::XCL, C::P2LO, C::P2HI>;
  constexpr int HB = C::HB, NV = C::NV, RU = C::RU, WIN = C::WIN, SR = C::SR, G = C::G, ZA = C::ZA;
  const kptr_t kargs = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  static_assert(SR % S == 0, "a wave keeps its row phase from step to step");
  __shared__ GT xs[C::NXR * C::XRG];
  __shared__ GT zs[C::NZR * C::ZRG];
  __shared__ GT cs[C::NCR * C::CRG];
  __shared__ GT hs[C::NHR];   // left halo columns of 2*lambda*w*r: one granule (the neighbour cell's last pixels) per row
  __shared__ double red[2][C::NW];

  double ostamp[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long tbase = SRMAP_EXP_MCLOCK ? __builtin_amdgcn_s_memtime() : 0ull;   // stamps are relative to the wave's entry
#define OSTAMP(k) do { if (SRMAP_EXP_MCLOCK) ostamp[k] = (double)(long long)(__builtin_amdgcn_s_memtime() - tbase); } while (0)
  OSTAMP(0);
  const int lane0 = threadIdx.x & 63;
  const int wv0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int lane = lane0, wv = wv0;
  const int strip = (int)blockIdx.x % A.m_nstrips, band = (int)blockIdx.x / A.m_nstrips;
  const int ch = blockIdx.y;
  const bool strip0 = strip == 0, lastst = strip == A.m_nstrips - 1;
  const bool edge_strip_c = strip0 || lastst;
  const int CJ0 = strip * C::CW, C0 = CJ0 * S;
  const int br0 = band * A.m_band_rows;   // multiples of SR; the plan admits H % SR == 0 only
  const int br1 = (br0 + A.m_band_rows < A.H) ? br0 + A.m_band_rows : A.H;
  const int nsteps = (br1 - br0) / SR;
  const size_t N = (size_t)A.W * A.H, nl = (size_t)A.wl * A.hl;
  const T* xplane = A.x + (size_t)ch * N;
  const T* ybase = A.y + (size_t)ch * nl;
  const bool want_data = (A.terms & SRMAP_TERM_DATA) != 0;
  const bool want_reg = REGK != 0 && (A.terms & SRMAP_TERM_REG) != 0;
  const T* wplane = (want_reg && A.w) ? A.w + (size_t)ch * N : nullptr;

  // ---- x-row requests: granule gi = 64 k + lane of a row -> (plane, cell).  With gi = 64 k + lane, plane = gi / XC,
  // cell = CJ0 - XCL + gi % XC the byte offset inside the row is an affine function of the lane per (k, plane): it is
  // re-derived from the lane at every request (a handful of 32-bit operations; kept in registers across the step the
  // offsets were the allocator's first spill candidates, and a spilled offset puts a scratch round trip in front of the
  // request).  Cells outside the image are skipped (their LDS granules were zeroed once: warp zero fill). ----
  auto req_geom = [&](int k, bool& ok, bool& cok, unsigned& vo) {
    const int gi = 64 * k + lane;
    const int plane = gi >= C::XC ? (gi >= 2 * C::XC ? 2 : 1) : 0;   // PL <= 2 planes (+ guard)
    const int cellpos = gi - plane * C::XC;
    const int cell = CJ0 - C::XCL + cellpos;
    ok = gi < C::XRG;
    cok = ok && (edge_strip_c ? (unsigned)cell < (unsigned)A.wl : true);
    vo = (unsigned)((cell * S + plane * G) * (int)sizeof(T));
  };
  // request HR row `row` of x into ring slot `slot` (rows outside the image read as zero: warp zero fill)
  auto request_row = [&](int row, int slot) {
    GT* dst = xs + slot * C::XRG;
    if ((unsigned)row < (unsigned)A.H) {  // uniform
      const T* rowp = xplane + (size_t)row * A.W;
#pragma unroll
      for (int k = 0; k < C::NLD; ++k) {
        bool ok, cok; unsigned vo;
        req_geom(k, ok, cok, vo);
        if (cok) dma16(vo, rowp, lds_addr(dst + 64 * k));
      }
    } else {
      int zi = 0;
      asm volatile("" : "+v"(zi));   // (a zero the compiler cannot keep in four registers across the whole kernel)
      GT zv;
#pragma unroll
      for (int e = 0; e < G; ++e) zv[e] = (T)zi;
#pragma unroll
      for (int k = 0; k < C::NLD; ++k) {
        bool ok, cok; unsigned vo;
        req_geom(k, ok, cok, vo);
        if (ok) dst[64 * k + lane] = zv;
      }
    }
  };
  // ---- fill: the first step's window; cells outside the image stay zero for the whole launch ----
  {
    GT zv;
#pragma unroll
    for (int e = 0; e < G; ++e) zv[e] = T(0);
    if (strip0 || lastst) {
      for (int q = wv; q < C::NXR; q += C::NW) {
#pragma unroll
        for (int k = 0; k < C::NLD; ++k) {
          bool ok, cok; unsigned vo;
          req_geom(k, ok, cok, vo);
          if (ok && !cok) xs[q * C::XRG + 64 * k + lane] = zv;
        }
      }
    }
    for (int q = wv; q < C::XWIN; q += C::NW) request_row(br0 - C::XLO + q, q);
    // first strip: the left halo columns of 2*lambda*w*r lie outside the image -- zero once, no column task
    if (threadIdx.x < C::NHR) hs[threadIdx.x] = zv;
  }

  // ---- per-wave constants: the row phase of the wave's residual row never changes (SR % S == 0) ----
  const int pr = (wv + ZA) % S;
  long long yoff[S];
  int io_bits = 0, jo_bits = 0;
#pragma unroll
  for (int pc = 0; pc < S; ++pc) {
    yoff[pc] = A.off0[pr][pc];
    io_bits |= (A.aux0[pr][pc].io & 1) << pc;
    jo_bits |= (A.aux0[pr][pc].jo & 1) << pc;
  }
  // residuals that do not exist, as bits over the thread's NV pixel slots: entries of LR row rc + 1 (io == 1), of LR cell
  // + 1 (jo == 1) among the own pixels; lane 0 of the first strip loses slot 0 when its entry has jo == 0 (LR column -1),
  // lane 63 of the last strip its last slot and the own slots with jo == 1 (LR column wl)
  int m_io1 = 0, m_jo1own = 0;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int pcv = v - HB, pc = posmod(pcv, S);
    m_io1 |= ((io_bits >> pc) & 1) << v;
    if (pcv >= 0 && pcv < S) m_jo1own |= ((jo_bits >> pc) & 1) << v;
  }
  const int c_lane0 = (strip0 && HB > 0 && ((jo_bits >> (S - 1)) & 1) == 0) ? 1 : 0;
  const int c_lane63 = lastst ? (m_jo1own | (HB > 0 ? (1 << (NV - 1)) : 0)) : 0;
  const bool edge_strip = strip0 || lastst;
  // carried input pointers of this wave: observations of the residual row (per column phase), weights of the regulariser
  // row; set for the virtual step's row br0 - SR + wv, advanced by SR rows per step
  const T* ypn[S];
  {
    const int zr = br0 - SR + wv + ZA;
    const int rc = (zr >= 0) ? zr / S : -((-zr + S - 1) / S);
#pragma unroll
    for (int pc = 0; pc < S; ++pc) ypn[pc] = ybase + (yoff[pc] + ((long long)rc * A.wl + CJ0));
  }
  const T* wrn = wplane != nullptr ? wplane + (ptrdiff_t)(br0 - SR + wv) * A.W : nullptr;
  const bool terms_all = want_data && want_reg;
  const int H = A.H;

  // Inputs of one wave row through registers: observations of residual row zrow (round 0 of the frame table), IRLS
  // weights of regulariser row gr.  FIXR: rows / strips at the image border -- entries without an LR row get any valid
  // address, element indices are clamped into the LR row (phase 1 masks those values).
  auto load_inputs = [&](int gr, bool do_z, bool do_r, bool fixr, T (&ypre)[NV], T (&wreg)[S]) {
    const int zrow = gr + ZA;
#pragma unroll
    for (int v = 0; v < NV; ++v) ypre[v] = T(0);
#pragma unroll
    for (int pc = 0; pc < S; ++pc) wreg[pc] = T(1);
    if (do_z) {
      if (!fixr) {
        // interior: the S row pointers of this wave (carried from step to step, advanced by a constant) + the lane
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int pcv = v - HB, pc = posmod(pcv, S), dc = floordiv(pcv, S);
          ypre[v] = (ypn[pc] + dc)[(unsigned)lane];
        }
      } else {
        const int rclass = zrow < 0 ? 1 : (zrow >= H ? 3 : (zrow >= H - S ? 2 : 0));
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int pcv = v - HB, pc = posmod(pcv, S), dc = floordiv(pcv, S);
          const int io = (io_bits >> pc) & 1, jo = (jo_bits >> pc) & 1;
          const bool ok_u = rclass == 0 || (rclass == 1 && io == 1) || (rclass == 2 && io == 0);
          const T* yp = ok_u ? ypn[pc] : ybase + (CJ0 + 1);   // no such LR row: any address valid under +-1
          const unsigned hi = lastst ? (unsigned)(63 - jo - (dc > 0 ? 1 : 0)) : 63u;   // uniform
          unsigned idx = (unsigned)lane < hi ? (unsigned)lane : hi;
          if (dc < 0) idx = (unsigned)lane + ((strip0 && jo == 0 && lane == 0) ? 1u : 0u);
          ypre[v] = (yp + dc)[idx];
        }
      }
    }
    if (do_r && wplane != nullptr && gr >= 0) {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) wreg[pc] = wrn[(unsigned)(C0 + S * lane) + pc];
    }
  };
  // advance the carried row pointers by `rows` HR rows (a multiple of S)
  auto advance_inputs = [&](int rows) {
#pragma unroll
    for (int pc = 0; pc < S; ++pc) ypn[pc] += (long long)(rows / S) * A.wl;
    if (wrn != nullptr) wrn += (size_t)rows * A.W;
  };
  auto make_ctl = [&](int gr, bool do_z, bool do_r) {
    P1Ctl ctl;
    const int zrow = gr + ZA;
    ctl.do_z = do_z; ctl.do_r = do_r;
    ctl.count_z = zrow >= br0 && zrow < br1;
    ctl.full = gr >= br0;
    const int allv = (1 << NV) - 1;
    const int ubad = zrow < 0 ? (allv & ~m_io1) : (zrow >= H ? allv : (zrow >= H - S ? m_io1 : 0));   // uniform
    ctl.badbits = ubad | (lane == 0 ? c_lane0 : 0) | (lane == 63 ? c_lane63 : 0);
    ctl.zero00 = gr == 0 && strip0;
    ctl.gr = gr; ctl.H = H;
    return ctl;
  };
  // left halo columns of 2*lambda*w*r (pixels of the neighbour strip's last cell: recomputed here), one row per lane:
  // rows rfirst + lane, lane < nrows, column C0 + COL (wave cw: COL = -1 - cw); into the halo ring hs
  auto col_task = [&](int cw, int rfirst, int nrows, T wcol) {
    if (lane < nrows) {
      const int q0 = rfirst - (br0 - C::XLO);            // ring numbering of the first row (uniform, >= 0)
      const int s0 = mwrap<C::NXR>(q0 % C::NXR + lane);  // nrows <= NXR
      int xe[WIN + 1];
#pragma unroll
      for (int i = 0; i <= WIN; ++i) xe[i] = mwrap<C::NXR>(s0 + i) * (C::XRG * G);
      const int hq = (rfirst - br0 + SR) % C::NHR + lane;  // halo ring: row r -> (r - br0 + SR) mod NHR; no wrap inside a task
      const T* xsT = reinterpret_cast<const T*>(xs);
      T* hsT = reinterpret_cast<T*>(hs);
      const KT K = load_konst<T, B, C::NP, ArgsT>(kargs);
      const int crow = rfirst + lane;
      const int ce = (hq < C::NHR ? hq : hq - C::NHR) * G;
      const bool cb = rfirst + nrows + WIN > K.H;
      if (cw == 0) {
        if (cb) m_halo_col<T, S, B, REGK, R, NW, -1, true>(K, xsT, hsT, wcol, xe, ce, crow, C0 - 1);
        else m_halo_col<T, S, B, REGK, R, NW, -1, false>(K, xsT, hsT, wcol, xe, ce, crow, C0 - 1);
      } else if (RU >= 2) {
        if (cb) m_halo_col<T, S, B, REGK, R, NW, (RU >= 2 ? -2 : -1), true>(K, xsT, hsT, wcol, xe, ce, crow, C0 - 2);
        else m_halo_col<T, S, B, REGK, R, NW, (RU >= 2 ? -2 : -1), false>(K, xsT, hsT, wcol, xe, ce, crow, C0 - 2);
      }
    }
  };
  const bool col_on = want_reg && RU > 0 && !strip0;

  T acc[S], zown[S];
  T ypre[NV], wreg[S];
  double cost = 0.0;
  OSTAMP(1);
  // ---- virtual step: halo rows of zh (br0 - HB .. br0 + ZA - 1) and of 2*lambda*w*r (br0 - RU .. br0 - 1) by the last
  // waves; halo columns of the first step's rows by waves 0 / 1.  Every input (these rows', step 0's) is requested
  // BEFORE the wait for the window: one memory round trip for the whole prologue ----
  {
    const int grv = br0 - SR + wv;
    const bool vz = want_data && wv >= SR - (HB + ZA), vr = want_reg && wv >= SR - RU;
    T ypv[NV], wrv[S];
    if (vz || vr) load_inputs(grv, vz, vr, true, ypv, wrv);
    T wcol = T(1);
    const bool colv = col_on && wv < RU;
    if (colv) {
      const int crow = br0 - RU + lane;
      if (lane < SR + RU && wplane != nullptr && crow >= 0) wcol = wplane[(size_t)crow * A.W + (C0 - 1 - wv)];
    }
    advance_inputs(SR);
    load_inputs(br0 + wv, want_data, want_reg, edge_strip || br0 + wv + zmax(WIN, S + ZA) >= H, ypre, wreg);
    // ---- border tasks: cost of the residuals whose z position lies OUTSIDE the image (they have no owner pixel;
    // ztile_dev.hpp, "Border blocks").  The pixels of the border frame are dealt to the middle half of each workgroup's waves, one
    // pixel per lane, and evaluated here, under the wait for the window (two memory round trips: table entry, then the
    // residual's B * B + 1 loads).  The plan admits non-positive frame offsets only: no in-image corrections. ----
    if (A.m_nbt > 0 && want_data && wv >= NW / 4 && wv < NW / 4 + NW / 2) {
      const int nwg = (int)gridDim.x;
      double cb = 0.0;
      for (int t = ((int)blockIdx.x * (NW / 2) + (wv - NW / 4)) * 64 + lane; t < A.m_nbt; t += nwg * (NW / 2) * 64) {
        int qr, qc;
        ring_pixel(t, A.W, H, A.ring, qr, qc);
        if ((unsigned)qr < (unsigned)H && (unsigned)qc < (unsigned)A.W) continue;  // (inside: nothing to correct under this plan)
        const int rc = dfdiv(qr, S), cc = dfdiv(qc, S);
        const ZEntry e = A.aux[(size_t)((qr - rc * S) * S + (qc - cc * S))];       // round 0 = the phase's only entry
        const int i = rc + e.io, j = cc + e.jo;
        if ((unsigned)i >= (unsigned)A.hl || (unsigned)j >= (unsigned)A.wl) continue;
        const int oy = e.oyx >> 16, ox = (int)(short)(e.oyx & 0xffff);
        const double r = (double)border_residual<T, S, B>(A, A.W, H, A.wl, xplane, ybase + (size_t)e.k * A.obs_C * nl, ox, oy, i, j);
        cb += r * r;
      }
      cost += (double)(S * S) * cb;
    }
    vm_wait_all();
    lds_barrier();
    OSTAMP(2);
    if (vz || vr) {
      pin(ypv);
      pin(wrv);
#pragma unroll
      for (int pc = 0; pc < S; ++pc) { acc[pc] = T(0); zown[pc] = T(0); }
      const T SCm = (lastst && lane == 63) ? T(0) : m_scale<T>();
      const GT* xb[WIN + 1];
      const int sxv = wv - SR + C::XLO;  // >= 0 for these waves
#pragma unroll
      for (int i = 0; i <= WIN; ++i) xb[i] = xs + (sxv + i) * C::XRG + lane;
      GT* zdst = zs + mwrapn<C::NZR>(wv - SR + ZA + HB) * C::ZRG + lane;
      GT* cdst = cs + mwrapn<C::NCR>(wv - SR + RU) * C::CRG + lane;
      const P1Ctl ctl = make_ctl(grv, vz, vr);
      const KT K = load_konst<T, B, C::NP, ArgsT>(kargs);
      W0T w0v;
      w0v.load(xb[0]);
      m_phase1<T, S, B, REGK, R, NW, true, true, false>(K, xb, zdst, cdst, w0v, ctl, lane, SCm, ypv, wrv, acc, zown, cost);
    }
    if (colv) col_task(wv, br0 - RU, SR + RU, wcol);
  }

  // ---- loop state: ring positions of this wave's rows, advanced by SR per step: x row r -> (16 n + wv + XLO) mod NXR,
  // zh row r + ZA -> (16 n + wv + ZA + HB) mod NZR, 2*lambda*w*r row r -> (16 n + wv + RU) mod NCR, halo ring row r ->
  // (16 n + wv + SR) mod NHR ----
  OSTAMP(3);
  int sx = wv + C::XLO, sz = (wv + ZA + HB) % C::NZR, sc = (wv + RU) % C::NCR, sh = (wv + SR) % C::NHR;
  int sxn = (C::XWIN + wv) % C::NXR;        // slot of the row this wave requests for the next step
  int nrow = br0 - C::XLO + C::XWIN + wv;   // and its HR row
  // What a phase needs beyond its inputs is prepared IN FRONT of the barrier that opens it -- LDS row addresses, the
  // phase's constants, the first window rows (x does not depend on the other waves' phase) -- under the arithmetic of
  // the waves still in the previous phase.  Behind a barrier all sixteen waves run the same code at once: whatever
  // stands there is paid sixteen times on the one scalar unit (profiles/r05_march_clock.txt: 0.7 .. 3.4 K cycles of
  // "head" per step before this reordering).
  const GT* xb[WIN + 1];
  GT* zdst;
  GT* cdst;
  W0T w0;
  auto prep_p1 = [&]() {
    asm volatile("" : "+v"(lane));   // lane-only values are re-derived per step (hoisted, they were the first spill candidates)
#pragma unroll
    for (int i = 0; i <= WIN; ++i) xb[i] = xs + mwrap<C::NXR>(sx + i) * C::XRG + lane;
    zdst = zs + sz * C::ZRG + lane;
    cdst = cs + sc * C::CRG + lane;
    w0.load(xb[0]);
  };
  prep_p1();
  lds_barrier();

  for (int n = 0; n < nsteps; ++n) {
    double stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    MSTAMP(0);
    const int gr = br0 + SR * n + wv;
    // the inputs requested at the end of the previous step are complete HERE (the compiler's own wait sits in front of
    // these pins); everything the step requests below is direct-to-LDS and waited for by hand
    pin(ypre);
    pin(wreg);
    // ---- head: the x rows step n + 1 adds to the window, straight into the ring slots step n - 1 released ----
    if (n + 1 < nsteps && !(SRMAP_EXP_MDBG & 16)) request_row(nrow, sxn);
    nrow += SR;
    sxn = mwrap<C::NXR>(sxn + SR);
    T wcol = T(1);   // weight of the halo-column pixel this lane evaluates under phase 2 (waves 0 / 1), next step's rows
    const bool col_next = col_on && wv < RU && n + 1 < nsteps;
    if (col_next && lane < SR && wplane != nullptr) wcol = wplane[(size_t)(gr - wv + SR + lane) * A.W + (C0 - 1 - wv)];
#pragma unroll
    for (int pc = 0; pc < S; ++pc) { acc[pc] = T(0); zown[pc] = T(0); }
    const bool need_fix = edge_strip || gr + zmax(WIN, S + ZA) >= H;
    MSTAMP(1);
    // ---- phase 1 ----
    {
      const T SCm = (lastst && lane == 63) ? T(0) : m_scale<T>();
      const KT K1 = load_konst<T, B, C::NP, ArgsT>(kargs);
      const bool row_fix = gr + zmax(WIN, S + ZA) >= H;   // the image's last rows
      if (terms_all && !need_fix) {
        P1Ctl ctl;
        ctl.count_z = gr + ZA < br1;
        m_phase1<T, S, B, REGK, R, NW, false, false, true>(K1, xb, zdst, cdst, w0, ctl, lane, SCm, ypre, wreg, acc, zown, cost);
      } else if (terms_all && !row_fix) {   // first / last strip, rows away from the bottom edge: column fix-ups only
        const P1Ctl ctl = make_ctl(gr, true, true);
        m_phase1<T, S, B, REGK, R, NW, true, false, true>(K1, xb, zdst, cdst, w0, ctl, lane, SCm, ypre, wreg, acc, zown, cost);
      } else {
        const P1Ctl ctl = make_ctl(gr, want_data, want_reg);
        m_phase1<T, S, B, REGK, R, NW, true, true, false>(K1, xb, zdst, cdst, w0, ctl, lane, SCm, ypre, wreg, acc, zown, cost);
      }
    }
    // ---- in front of barrier 1: phase 2's addresses and the row of x it starts with ----
    const GT* xb2[RU + 1];
    const GT* cb2[RU + 1];
    const GT* hb2[RU + 1];
    const GT* zb2[B];
#pragma unroll
    for (int i = 0; i <= RU; ++i) {
      xb2[i] = xs + mwrapn<C::NXR>(sx - i) * C::XRG + lane;
      cb2[i] = cs + mwrapn<C::NCR>(sc - i) * C::CRG + lane;
      hb2[i] = hs + mwrapn<C::NHR>(sh - i);
    }
#pragma unroll
    for (int a = 0; a < B; ++a) zb2[a] = zs + mwrapn<C::NZR>(sz - ZA - HB + a) * C::ZRG + lane;  // zh row gr - HB + a
    W2T xw0;
    xw0.load(xb2[0]);
    MSTAMP(2);
    vm_wait_all();   // the x rows requested at the head have landed long since (and the halo-column weight)
    MSTAMP(3);
    lds_barrier();
    MSTAMP(4);

    // ---- phase 2 ----
    {
      const KT K2 = load_konst<T, B, C::NP, ArgsT>(kargs);
      // the NEXT step's inputs are requested in the MIDDLE of phase 2 (behind its data part): phase 1 has consumed this
      // step's, the registers are free, the waves of a SIMD pass this point one after the other (80 KB of loads per step
      // are ~1.3 K cycles of the CU's vector-memory path), and the loads have the rest of the step to land
      m_phase2<T, S, B, REGK, R, NW>(K2, xb2, cb2, hb2, zb2, xw0, lane, want_data, want_reg, zown, acc, [&]() {
        advance_inputs(SR);
        if (n + 1 < nsteps) load_inputs(gr + SR, want_data, want_reg, edge_strip || gr + SR + zmax(WIN, S + ZA) >= H, ypre, wreg);
      });
      T* dst = A.g + (size_t)ch * N + (size_t)gr * A.W;
      MSTAMP(5);
      if (SRMAP_EXP_MCLOCK) {
        if (lane == 0) { acc[0] = (T)stamp[0]; acc[1] = (T)stamp[1]; acc[2] = (T)stamp[2]; acc[3] = (T)stamp[3]; }
        if (lane == 1) { acc[0] = (T)stamp[4]; acc[1] = (T)stamp[5]; }
      }
#pragma unroll
      for (int pc = 0; pc < S; ++pc) __builtin_nontemporal_store(acc[pc], &dst[(unsigned)(C0 + S * lane) + pc]);
    }
    // under the other waves' phase 2 (waves 0 / 1 finish theirs first): halo columns of the NEXT step's rows, this
    // wave's inputs and phase-1 preparation of the next step -- all in front of the barrier, where the workgroup's
    // arithmetic covers them
    if (col_next) col_task(wv, gr - wv + SR, SR, wcol);
    sx = mwrap<C::NXR>(sx + SR);
    sz = (sz + SR) % C::NZR;   // uniform: scalar arithmetic
    sc = (sc + SR) % C::NCR;
    sh = (sh + SR) % C::NHR;
    if (n + 1 < nsteps) prep_p1();
    lds_barrier();
  }

  OSTAMP(4);
  if (SRMAP_EXP_MCLOCK && lane == 2) {
    T* dst = A.g + (size_t)ch * N + (size_t)(br0 + wv) * A.W + (unsigned)(C0 + S * lane);
    dst[0] = (T)ostamp[0]; dst[1] = (T)ostamp[1]; dst[2] = (T)ostamp[2]; dst[3] = (T)ostamp[3];
    dst[4] = (T)ostamp[4];
  }
  // ---- cost partial of this workgroup ----
  {
    const double cw = wave_sum_d(cost);
    if (lane == 0) red[0][wv] = cw;
    __syncthreads();
    if (threadIdx.x == 0) {
      double c = 0.0;
#pragma unroll
      for (int i = 0; i < C::NW; ++i) c += red[0][i];
      const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
      put_partial<false>(A, b, c, 0.0);
    }
    if (A.mfinish && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1) {
      __syncthreads();
      finish_block<false, C::NT>(A, &red[0][0]);
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// host side

// waves per workgroup and resident workgroups per CU of the instance a dtype runs: f64 fills the LDS with one 16-wave
// workgroup (152 KB of rings); f32 rings are half as wide, and two independent 8-wave workgroups -- each other's cover
// for barrier and request waits, which a single workgroup has nothing to fill with -- fit with room to spare
#ifndef SRMAP_EXP_MNW32
#define SRMAP_EXP_MNW32 8
#endif
template <typename T> struct MShape { static constexpr int NW = SRMAP_EXP_MTILE ? 8 : (sizeof(T) == 8 ? 16 : SRMAP_EXP_MNW32); static constexpr int WG_PER_CU = (NW == 16 ? 1 : SRMAP_EXP_MWG8); };

template <typename T>
bool zmarch_covers(const srmap_problem* p, const Geometry& geo, const ZPlan& z, int regk, int regr, unsigned terms,
                   const T* g, const T* dvec, int* nstrips, int* band_rows) {
  if (z.subpix || dvec != nullptr || g == nullptr) return false;
  if (geo.s != 4 || geo.b != 3 || regk != 2 || regr != 3) return false;
  if (p->ov_hook != nullptr) return false;
  if (geo.rr0 != 0 || geo.rr1 != geo.H || geo.cr0 != 0 || geo.cr1 != geo.H) return false;
  constexpr int SR = MShape<T>::NW;
  if (geo.W % (64 * geo.s) != 0 || geo.H % SR != 0) return false;
  // every pixel phase owns exactly one residual (K = S * S frames on distinct phases), frame offsets in [-(S-1), 0]:
  // the image border then needs the selects of the FIX instances only (zmarch_dev.hpp)
  if (z.MS != 1) return false;
  for (int pr = 0; pr < geo.s; ++pr)
    for (int pc = 0; pc < geo.s; ++pc) {
      if (z.h_cnt[pr * 8 + pc] != 1) return false;
      const ZEntry& e = z.h_aux0[pr * 4 + pc];
      const int oy = e.oyx >> 16, ox = (int)(short)(e.oyx & 0xffff);
      if (oy > 0 || ox > 0 || oy <= -geo.s || ox <= -geo.s) return false;
      if (e.io < 0 || e.io > 1 || e.jo < 0 || e.jo > 1) return false;
    }
  if (p->ctx->num_cus <= 0) return false;
  (void)terms;
  const int ns = geo.w / 64;
  // bands: as many workgroups as the chip holds at once (resident for the whole launch), rows in multiples of the step
  int nb = p->ctx->num_cus * MShape<T>::WG_PER_CU / ns;
  if (nb < 1) nb = 1;
  int rows = (geo.H + nb - 1) / nb;
  rows = (rows + SR - 1) / SR * SR;
  if (rows < 2 * SR) rows = 2 * SR;   // small images: fewer workgroups than the chip holds (a band amortises its prologue)
  if (SRMAP_EXP_MTILE) rows = SR;
  if (rows > geo.H) rows = geo.H;
  *nstrips = ns; *band_rows = rows;
  return true;
}

template <typename T, int S, int B, int REGK, int R>
static int launch_m(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g, const T* wts,
                    const ZPlan& z, double* partials, int* nblocks, hipStream_t st, MFin mfin, int nstrips, int band_rows, int nbord) {
  using ZC = ZCfg<T, S, B, REGK, R>;
  constexpr int NW = MShape<T>::NW;
  ZArgs<T, B, ZC::NP> A;
  fill_zargs<T, S, B, REGK, R>(A, p, geo, obs_c0, terms, x, g, wts, z, partials, (const T*)nullptr, (double*)nullptr);
  const int nbands = (geo.H + band_rows - 1) / band_rows;
  dim3 grid(nstrips * nbands, geo.C, 1);
  A.m_nstrips = nstrips; A.m_band_rows = band_rows;
  A.m_nbt = ((terms & SRMAP_TERM_DATA) && z.n_ring > 0) ? z.n_ring : 0;
  A.bd = (const BorderArgs<T>*)z.d_bd;
  A.rbuf = nullptr; A.spw = z.d_spw; A.Dr = z.Dr;
  A.mpart = z.d_mpart;
  A.mpart_gd = z.d_mpart ? z.d_mpart + z.mpart_cap : nullptr;
  A.cost_out = p->d_cost;
  A.tag_slot = p->eval_pub_tag_slot;
  A.tag = p->eval_pub_tag;
  A.sel_mode = 0; A.sel0 = 0; A.sel1 = 0;
  A.nby = 0;
  A.n_tile_partials = (int)(grid.x * grid.y);
  A.mfinish = mfin.on ? 1 : 0;
  A.n_partials = A.n_tile_partials + nbord;
  A.pub = nullptr;
  A.xpart = mfin.xpart; A.n_xpart = mfin.n_xpart;
  if (mfin.on && (size_t)A.n_partials > z.mpart_cap) return set_error(p->ctx, SRMAP_EHIP, "granule capacity");
  hipLaunchKernelGGL((k_eval_m<T, S, B, REGK, R, NW, false>), grid, dim3(64 * NW), 0, st, A);
  *nblocks = A.n_tile_partials;
  SRMAP_HIP(p->ctx, hipGetLastError());
  return SRMAP_OK;
}

template <typename T>
int launch_zmarch(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g, const T* wts,
                  const ZPlan& z, int regk, int regr, double* partials, int* nblocks, hipStream_t st, MFin mfin,
                  int nstrips, int band_rows, int nbord) {
  (void)regk; (void)regr;
  return launch_m<T, 4, 3, 2, 3>(p, geo, obs_c0, terms, x, g, wts, z, partials, nblocks, st, mfin, nstrips, band_rows, nbord);
}
template int launch_zmarch<float>(srmap_problem*, const Geometry&, int, unsigned, const float*, float*, const float*,
                                  const ZPlan&, int, int, double*, int*, hipStream_t, MFin, int, int, int);
template int launch_zmarch<double>(srmap_problem*, const Geometry&, int, unsigned, const double*, double*, const double*,
                                   const ZPlan&, int, int, double*, int*, hipStream_t, MFin, int, int, int);
template bool zmarch_covers<float>(const srmap_problem*, const Geometry&, const ZPlan&, int, int, unsigned, const float*,
                                   const float*, int*, int*);
template bool zmarch_covers<double>(const srmap_problem*, const Geometry&, const ZPlan&, int, int, unsigned, const double*,
                                    const double*, int*, int*);

void zmarch_preload() {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_eval_m<double, 4, 3, 2, 3, MShape<double>::NW, false>));
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_eval_m<float, 4, 3, 2, 3, MShape<float>::NW, false>));
}

}  // namespace srmap
