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

namespace srmap {

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(lds_ptr_t)p; }
// Direct-to-LDS request: every active lane's 16 bytes at base + voff land at lds + lane * 16.  Inline assembly on
// purpose: the compiler's builtin brackets the request with s_waitcnt vmcnt(0) and waits for it in front of every
// barrier; here the one wait is placed by hand in front of the step's first barrier.
__device__ __forceinline__ void dma16(unsigned voff, const void* base, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void vm_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// workgroup barrier for LDS traffic only (no release fence over global memory: the g stores stay in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T, int S, int B, int REGK, int R, bool WD>
__global__ __launch_bounds__(1024, 1) void k_eval_m(ZArgs<T, B, ZCfg<T, S, B, REGK, R>::NP> A) {
  using C = MCfg<T, S, B, REGK, R>;
  using GT = typename Gran<T, C::G>::type;
  constexpr int HB = C::HB, NV = C::NV, RU = C::RU, WIN = C::WIN, SR = C::SR, G = C::G, ZA = C::ZA;
  static_assert(SR % S == 0, "a wave keeps its row phase from step to step");
  __shared__ GT xs[C::NXR * C::XRG];
  __shared__ GT zs[C::NZR * C::ZRG];
  __shared__ GT cs[C::NCR * C::CRG];
  __shared__ double red[2][C::NW];

  const int lane0 = threadIdx.x & 63;
  const int wv0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int lane = lane0, wv = wv0;
  const int strip = (int)blockIdx.x % A.m_nstrips, band = (int)blockIdx.x / A.m_nstrips;
  const int ch = blockIdx.y;
  const bool strip0 = strip == 0, lastst = strip == A.m_nstrips - 1;
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

  // ---- x-row requests: granule gi = 64 k + lane of a row -> (plane, cell).  The per-lane byte offsets are re-derived from
  // the lane's own column at every request (a few 32-bit operations per step; kept in registers across the step they were
  // the allocator's first spill candidates, and a spilled offset puts a scratch round trip in front of the request) ----
  auto req_geom = [&](int k, bool& ok, bool& cok, unsigned& vo) {
    const int gi = 64 * k + lane;
    const int plane = gi >= C::XC ? (gi >= 2 * C::XC ? 2 : 1) : 0;   // PL <= 2 planes (+ guard)
    const int cellpos = gi - plane * C::XC;
    const int cell = CJ0 - C::XCL + cellpos;
    ok = gi < C::XRG;
    cok = ok && (unsigned)cell < (unsigned)A.wl;
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
    if (REGK != 0 && strip0 && threadIdx.x < C::NCR * C::PL) cs[(threadIdx.x / C::PL) * C::CRG + (threadIdx.x % C::PL) * C::CC] = zv;
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
  T acc[S], zown[S];
  double cost_data = 0.0, cost_reg = 0.0;
  vm_wait_all();
  lds_barrier();

  // ring positions of this wave's rows, advanced by SR per step: x row r -> (16 n + wv + XLO) mod NXR, zh row r + ZA
  // -> (16 n + wv + ZA + HB) mod NZR, 2*lambda*w*r row r -> (16 n + wv + RU) mod NCR
  int sx = mwrapn<C::NXR>(wv - SR + C::XLO), sz = mwrapn<C::NZR>(wv - SR + ZA + HB), sc = mwrapn<C::NCR>(wv - SR + RU);
  int sxn = (C::XWIN + wv) % C::NXR;        // slot of the row this wave requests for the next step
  int nrow = br0 - C::XLO + C::XWIN + wv;   // and its HR row
  const bool terms_all = want_data && want_reg;

  for (int n = -1; n < nsteps; ++n) {
    // Everything below that depends only on the lane / the wave index is RE-DERIVED per step: the values are made
    // opaque here so that the compiler cannot hoist those computations out of the loop -- hoisted, they became dozens of
    // loop-carried registers, the allocator's spill candidates (scratch round trips in front of the requests,
    // v_writelane / v_readlane chains at the head of every step).
    asm volatile("" : "+v"(lane));
    asm volatile("" : "+s"(wv));
    const unsigned gcoff = (unsigned)(C0 + S * lane);
    const int r0 = br0 + SR * n;
    const int gr = r0 + wv, zrow = gr + ZA;
    const bool virt = n < 0;
    // ---- head: next step's new x rows straight into LDS; this step's observations and weights into registers ----
    if (!virt) {
      if (n + 1 < nsteps && !(SRMAP_EXP_MDBG & 16)) request_row(nrow, sxn);
      nrow += SR;
      sxn = mwrap<C::NXR>(sxn + SR);
    }
    const T SCm = (lastst && lane == 63) ? T(0) : m_scale<T>();
    P1Ctl ctl;
    // residual rows br0 - HB .. br1 - 1 + HB, regulariser rows br0 - RU .. br1 - 1
    ctl.do_z = want_data && (!virt || wv >= SR - (HB + ZA));
    ctl.do_r = want_reg && (!virt || wv >= SR - RU);
    ctl.count_z = zrow >= br0 && zrow < br1;
    ctl.full = !virt;
    ctl.rclass = zrow < 0 ? 1 : (zrow >= A.H ? 3 : (zrow >= A.H - S ? 2 : 0));
    ctl.io_bits = io_bits; ctl.jo_bits = jo_bits;
    ctl.strip0 = strip0; ctl.lastst = lastst;
    ctl.gr = gr; ctl.H = A.H;
    const bool need_fix = strip0 || lastst || ctl.rclass != 0 || gr < 0 || gr + WIN >= A.H;
    T ypre[NV];
    T dbgv[S] = {};
#pragma unroll
    for (int v = 0; v < NV; ++v) ypre[v] = T(0);
    if (ctl.do_z && !(SRMAP_EXP_MDBG & 1)) {
      const int rc = (zrow >= 0) ? zrow / S : -((-zrow + S - 1) / S);
      const long long yrow = (long long)rc * A.wl + CJ0;
      const T* yp[S];
#pragma unroll
      for (int pc = 0; pc < S; ++pc) {
        const int io = (io_bits >> pc) & 1;
        const bool ok_u = ctl.rclass == 0 || (ctl.rclass == 1 && io == 1) || (ctl.rclass == 2 && io == 0);
        yp[pc] = (ok_u && !(SRMAP_EXP_MDBG & 32)) ? ybase + (yoff[pc] + yrow) : ybase + (CJ0 + 1);   // no such LR row: any address that stays valid under the +-1 below
        if (SRMAP_EXP_MDBG & 128) dbgv[pc] = (T)(yoff[pc] + yrow) + (ok_u ? T(0) : T(0.5));   // an entry without LR row: any valid address
      }
      // every address stays inside its LR row [0, wl): the element index lane + dc is clamped to [-jo, wl - 1 - jo - CJ0]
      // (the values of clamped lanes are masked in phase 1: no such LR pixel)
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int pcv = v - HB, pc = posmod(pcv, S), dc = floordiv(pcv, S);
        const int jo = (jo_bits >> pc) & 1;
        const unsigned hi = lastst ? (unsigned)(63 - jo - (dc > 0 ? 1 : 0)) : 63u;   // uniform
        unsigned idx = (unsigned)lane < hi ? (unsigned)lane : hi;
        if (dc < 0) idx = (unsigned)lane + ((strip0 && jo == 0 && lane == 0) ? 1u : 0u);
        ypre[v] = (SRMAP_EXP_MDBG & 64) ? yp[pc][(unsigned)lane] : (yp[pc] + dc)[idx];
      }
    }
    T wreg[S];
#pragma unroll
    for (int pc = 0; pc < S; ++pc) wreg[pc] = T(1);
    if (ctl.do_r && wplane != nullptr && gr >= 0 && !(SRMAP_EXP_MDBG & 2)) {
      const T* wrow = wplane + (size_t)gr * A.W;
#pragma unroll
      for (int pc = 0; pc < S; ++pc) wreg[pc] = wrow[gcoff + pc];
    }
    // left halo columns of 2*lambda*w*r: waves 4 / 5 (columns -1 / -2), one row per lane
    const bool col_task = !(SRMAP_EXP_MDBG & 4) && want_reg && RU > 0 && !strip0 && (wv == 4 || wv == 5) && (wv - 4) < RU;
    T wcol = T(1);
    const int crow = r0 + lane;  // lane < SR: row of this step (virtual step: only its last RU rows)
    const bool col_lane = col_task && lane < SR && (!virt || lane >= SR - RU);
    if (col_lane && wplane != nullptr && crow >= 0) wcol = wplane[(size_t)crow * A.W + (C0 - (wv == 4 ? 1 : 2))];
#pragma unroll
    for (int pc = 0; pc < S; ++pc) { acc[pc] = T(0); zown[pc] = T(0); }

    // ---- phase 1 ----
    if (ctl.do_z || ctl.do_r) {
      const GT* xb[WIN + 1];
#pragma unroll
      for (int i = 0; i <= WIN; ++i) xb[i] = xs + mwrap<C::NXR>(sx + i) * C::XRG + lane;
      GT* zdst = zs + sz * C::ZRG + lane;
      GT* cdst = cs + sc * C::CRG + lane;
#ifndef SRMAP_EXP_MVAR
#define SRMAP_EXP_MVAR 7
#endif
      if ((SRMAP_EXP_MVAR & 1) && !virt && terms_all && !need_fix)
        m_phase1<T, S, B, REGK, R, false, true>(A, xb, zdst, cdst, ctl, lane, SCm, ypre, wreg, acc, zown, cost_data, cost_reg);
      else if ((SRMAP_EXP_MVAR & 2) && !virt && terms_all)
        m_phase1<T, S, B, REGK, R, true, true>(A, xb, zdst, cdst, ctl, lane, SCm, ypre, wreg, acc, zown, cost_data, cost_reg);
      else if (SRMAP_EXP_MVAR & 4)
        m_phase1<T, S, B, REGK, R, true, false>(A, xb, zdst, cdst, ctl, lane, SCm, ypre, wreg, acc, zown, cost_data, cost_reg);
    }
    if (col_lane) {
      // per-lane slots of rows crow .. crow + WIN and of the 2*lambda*w*r row (uniform base + lane, one wrap each)
      int xe[WIN + 1];
      const int sxb = mwrapn<C::NXR>(sx - wv), scb = mwrapn<C::NCR>(sc - wv);  // slots of the step's first row
      int s0 = mwrap<C::NXR>(sxb + lane);
#pragma unroll
      for (int i = 0; i <= WIN; ++i) xe[i] = mwrap<C::NXR>(s0 + i) * (C::XRG * G);
      const int ce = mwrap<C::NCR>(scb + lane) * (C::CRG * G);
      const T* xsT = reinterpret_cast<const T*>(xs);
      T* csT = reinterpret_cast<T*>(cs);
      const bool cb = r0 + SR + WIN > A.H;
      if (wv == 4) {
        if (cb) m_halo_col<T, S, B, REGK, R, -1, true>(A, xsT, csT, wcol, xe, ce, crow, C0 - 1);
        else m_halo_col<T, S, B, REGK, R, -1, false>(A, xsT, csT, wcol, xe, ce, crow, C0 - 1);
      } else if (RU >= 2) {
        if (cb) m_halo_col<T, S, B, REGK, R, (RU >= 2 ? -2 : -1), true>(A, xsT, csT, wcol, xe, ce, crow, C0 - 2);
        else m_halo_col<T, S, B, REGK, R, (RU >= 2 ? -2 : -1), false>(A, xsT, csT, wcol, xe, ce, crow, C0 - 2);
      }
    }
    vm_wait_all();   // this step's inputs are consumed; the x rows requested at the head have landed long since
    lds_barrier();

    // ---- phase 2 ----
    if (!virt) {
      const GT* xb2[RU + 1];
      const GT* cb2[RU + 1];
      const GT* zb2[B];
#pragma unroll
      for (int i = 0; i <= RU; ++i) {
        xb2[i] = xs + mwrapn<C::NXR>(sx - i) * C::XRG + lane;
        cb2[i] = cs + mwrapn<C::NCR>(sc - i) * C::CRG + lane;
      }
#pragma unroll
      for (int a = 0; a < B; ++a) zb2[a] = zs + mwrapn<C::NZR>(sz - ZA - HB + a) * C::ZRG + lane;  // zh row gr - HB + a
      m_phase2<T, S, B, REGK, R>(A, xb2, cb2, zb2, want_data, want_reg, zown, acc);
      T* dst = A.g + (size_t)ch * N + (size_t)gr * A.W;
#pragma unroll
      for (int pc = 0; pc < S; ++pc) if (!(SRMAP_EXP_MDBG & 8)) __builtin_nontemporal_store((SRMAP_EXP_MDBG & 128) ? dbgv[pc] : acc[pc], &dst[gcoff + pc]);
      lds_barrier();
    }
    sx = mwrap<C::NXR>(sx + SR);
    sz = (sz + SR) % C::NZR;   // uniform: scalar arithmetic
    sc = (sc + SR) % C::NCR;
  }

  // ---- cost partial of this workgroup ----
  {
    const double cw = wave_sum_d((double)(S * S) * cost_data + cost_reg);
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

template <typename T>
bool zmarch_covers(const srmap_problem* p, const Geometry& geo, const ZPlan& z, int regk, int regr, unsigned terms,
                   const T* g, const T* dvec, int* nstrips, int* band_rows) {
  if (z.subpix || dvec != nullptr || g == nullptr) return false;
  if (sizeof(T) != 8) return false;
  if (geo.s != 4 || geo.b != 3 || regk != 2 || regr != 3) return false;
  if (p->ov_hook != nullptr) return false;
  if (geo.rr0 != 0 || geo.rr1 != geo.H || geo.cr0 != 0 || geo.cr1 != geo.H) return false;
  if (geo.W % (64 * geo.s) != 0 || geo.H % 16 != 0) return false;
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
  // bands: as many workgroups as CUs (one resident workgroup each), rows in multiples of the step
  int nb = p->ctx->num_cus / ns;
  if (nb < 1) nb = 1;
  int rows = (geo.H + nb - 1) / nb;
  rows = (rows + 15) / 16 * 16;
  if (rows < 32) return false;
  *nstrips = ns; *band_rows = rows;
  return true;
}

template <typename T, int S, int B, int REGK, int R>
static int launch_m(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g, const T* wts,
                    const ZPlan& z, double* partials, int* nblocks, hipStream_t st, MFin mfin, int nstrips, int band_rows, int nbord) {
  using ZC = ZCfg<T, S, B, REGK, R>;
  ZArgs<T, B, ZC::NP> A;
  fill_zargs<T, S, B, REGK, R>(A, p, geo, obs_c0, terms, x, g, wts, z, partials, (const T*)nullptr, (double*)nullptr);
  const int nbands = (geo.H + band_rows - 1) / band_rows;
  dim3 grid(nstrips * nbands, geo.C, 1);
  A.m_nstrips = nstrips; A.m_band_rows = band_rows;
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
  A.n_partials = A.n_tile_partials + nbord;   // border blocks of the launch in front publish behind the workgroups
  A.pub = nullptr;
  A.xpart = mfin.xpart; A.n_xpart = mfin.n_xpart;
  if (mfin.on && (size_t)A.n_partials > z.mpart_cap) return set_error(p->ctx, SRMAP_EHIP, "granule capacity");
  hipLaunchKernelGGL((k_eval_m<T, S, B, REGK, R, false>), grid, dim3(1024), 0, st, A);
  *nblocks = A.n_tile_partials;
  SRMAP_HIP(p->ctx, hipGetLastError());
  return SRMAP_OK;
}

template <typename T>
int launch_zmarch(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g, const T* wts,
                  const ZPlan& z, int regk, int regr, double* partials, int* nblocks, hipStream_t st, MFin mfin,
                  int nstrips, int band_rows, int nbord) {
  (void)regk; (void)regr;
  return set_error(p->ctx, SRMAP_EUNSUPPORTED, "no marching kernel for this dtype");
}
template <>
int launch_zmarch<double>(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const double* x, double* g,
                          const double* wts, const ZPlan& z, int regk, int regr, double* partials, int* nblocks,
                          hipStream_t st, MFin mfin, int nstrips, int band_rows, int nbord) {
  (void)regk; (void)regr;
  return launch_m<double, 4, 3, 2, 3>(p, geo, obs_c0, terms, x, g, wts, z, partials, nblocks, st, mfin, nstrips, band_rows, nbord);
}
template int launch_zmarch<float>(srmap_problem*, const Geometry&, int, unsigned, const float*, float*, const float*,
                                  const ZPlan&, int, int, double*, int*, hipStream_t, MFin, int, int, int);
template bool zmarch_covers<float>(const srmap_problem*, const Geometry&, const ZPlan&, int, int, unsigned, const float*,
                                   const float*, int*, int*);
template bool zmarch_covers<double>(const srmap_problem*, const Geometry&, const ZPlan&, int, int, unsigned, const double*,
                                    const double*, int*, int*);

void zmarch_preload() {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_eval_m<double, 4, 3, 2, 3, false>));
}

}  // namespace srmap
