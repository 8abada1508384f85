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

template <typename T, int S, int B, int REGK, int R, bool WD>
__global__ __launch_bounds__(1024, 1) void k_eval_m(ZArgs<T, B, ZCfg<T, S, B, REGK, R>::NP> A) {
  using C = MCfg<T, S, B, REGK, R>;
  using ZC = typename C::Z;
  using GT = typename Gran<T, C::G>::type;
  constexpr int HB = C::HB, NV = C::NV, RU = C::RU, WIN = C::WIN, SR = C::SR, G = C::G, ZA = C::ZA;
  __shared__ GT xs[C::NXR * C::XRG];
  __shared__ GT zs[C::NZR * C::ZRG];
  __shared__ GT cs[C::NCR * C::CRG];
  __shared__ double red[2][C::NW];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = (int)blockIdx.x % A.m_nstrips, band = (int)blockIdx.x / A.m_nstrips;
  const int ch = blockIdx.y;
  const int CJ0 = strip * C::CW, C0 = CJ0 * S;
  const int br0 = band * A.m_band_rows;
  const int br1 = (br0 + A.m_band_rows < A.H) ? br0 + A.m_band_rows : A.H;
  const int nsteps = (br1 - br0 + SR - 1) / SR;
  const size_t N = (size_t)A.W * A.H, nl = (size_t)A.wl * A.hl;
  const T* xplane = A.x + (size_t)ch * N;
  const T* ybase = A.y + (size_t)ch * nl;
  const bool want_data = (A.terms & SRMAP_TERM_DATA) != 0;
  const bool want_reg = REGK != 0 && (A.terms & SRMAP_TERM_REG) != 0;
  const T* wplane = (want_reg && A.w) ? A.w + (size_t)ch * N : nullptr;
  const int gc0 = C0 + S * lane;

  // ---- per-lane constants of the x-row requests: granule gi = 64 k + lane of a row -> (plane, cell) ----
  unsigned voff[C::NLD];
  bool gok[C::NLD], colok[C::NLD];
#pragma unroll
  for (int k = 0; k < C::NLD; ++k) {
    const int gi = 64 * k + lane;
    const int plane = gi / C::XC, cellpos = gi - plane * C::XC;
    const int cell = CJ0 - C::XCL + cellpos;
    gok[k] = gi < C::XRG;
    colok[k] = gok[k] && (unsigned)cell < (unsigned)A.wl;
    voff[k] = (unsigned)((cell * S + plane * G) * (int)sizeof(T));
  }
  // request x row q of the band's ring numbering (HR row br0 - XLO + q) into its slot
  auto request_row = [&](int q, int slot) {
    const int row = br0 - C::XLO + q;
    GT* dst = xs + slot * C::XRG;
    if ((unsigned)row < (unsigned)A.H) {  // uniform
      const T* rowp = xplane + (size_t)row * A.W;
#pragma unroll
      for (int k = 0; k < C::NLD; ++k)
        if (colok[k]) dma16(voff[k], rowp, lds_addr(dst + 64 * k));
    } else {  // rows outside the image read as zero (warp zero fill)
      GT zv;
#pragma unroll
      for (int e = 0; e < G; ++e) zv[e] = T(0);
#pragma unroll
      for (int k = 0; k < C::NLD; ++k)
        if (gok[k]) dst[64 * k + lane] = zv;
    }
  };
  // ---- fill: the first step's window; cells outside the image stay zero for the whole launch ----
  {
    GT zv;
#pragma unroll
    for (int e = 0; e < G; ++e) zv[e] = T(0);
    for (int q = wv; q < C::NXR; q += C::NW) {
#pragma unroll
      for (int k = 0; k < C::NLD; ++k)
        if (gok[k] && !colok[k]) xs[q * C::XRG + 64 * k + lane] = zv;
    }
    for (int q = wv; q < C::XWIN; q += C::NW) request_row(q, q);
  }

  T acc[S], zown[S];
  double cost_data = 0.0, cost_reg = 0.0, gdsum = 0.0;
  (void)gdsum;
  vm_wait_all();
  lds_barrier();

  // ring positions of this wave's rows, advanced by SR per step: x row r -> (16 n + wv + XLO) mod NXR, zh row r + ZA
  // -> (16 n + wv + ZA + HB) mod NZR, 2*lambda*w*r row r -> (16 n + wv + RU) mod NCR
  int sx = wv - SR + C::XLO, sz = wv - SR + ZA + HB, sc = wv - SR + RU;
  sx = sx < 0 ? sx + C::NXR : sx;
  sz = sz < 0 ? sz + C::NZR : sz;
  sc = sc < 0 ? sc + C::NCR : sc;

  const int rm = A.E + HB + 1, cm = (A.E + HB + S) / S + 1;
  const bool col_edge = (CJ0 - cm < 0) || (CJ0 + C::CW + cm > A.wl) || A.cr0 > 0 || A.cr1 < A.H;
  const bool col_border = C0 + C::TW + WIN > A.W;

  for (int n = -1; n < nsteps; ++n) {
    const int r0 = br0 + SR * n;
    const int gr = r0 + wv, zrow = gr + ZA;
    const bool virt = n < 0;
    const bool p1_on = !virt || wv >= SR - C::VW;
    // ---- head: next step's new x rows straight into LDS; this step's observations and weights into registers ----
    if (!virt && n + 1 < nsteps) {
      const int q = SR * (n + 1) + C::XWIN - SR + wv;  // new row wv of step n + 1: q = 16 (n + 1) + XWIN - 16 + wv
      request_row(q, q % C::NXR);
    }
    // residual rows br0 - HB .. br1 - 1 + HB, regulariser rows br0 - RU .. br1 - 1
    const bool do_z = want_data && p1_on && (!virt || wv >= SR - (HB + ZA)) && zrow <= br1 - 1 + HB;
    const bool do_r = want_reg && p1_on && (!virt || wv >= SR - RU) && gr < br1;
    const bool count_z = zrow >= br0 && zrow < br1;
    const bool full = gr >= br0 && gr < br1;
    const bool z_slow = (zrow - rm < 0) || (zrow + 1 + rm > A.H) || col_edge;
    const bool r_slow = (gr + 1 + WIN > A.H) || col_border || gr < 0;
#ifdef SRMAP_EXP_MFAST   // TIMING ONLY: every row through the interior path (wrong at the image border)
    const bool slow = false; (void)z_slow; (void)r_slow;
#else
    const bool slow = z_slow || r_slow;
#endif
    T ypre[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) ypre[v] = T(0);
    if (do_z) {
      int rc, pr;
      row_phase<S>(zrow, rc, pr);
      int cn[S];
#pragma unroll
      for (int pc = 0; pc < S; ++pc) cn[pc] = A.cntk[pr][pc];
      if (z_slow || slow) load_obs_row<T, S, ZC, true>(A, pr, rc, 0, CJ0, lane, ybase, cn, ypre);
      else load_obs_row<T, S, ZC, false>(A, pr, rc, 0, CJ0, lane, ybase, cn, ypre);
    }
    T wreg[S];
#pragma unroll
    for (int pc = 0; pc < S; ++pc) wreg[pc] = T(1);
    if (do_r && wplane != nullptr && (unsigned)gr < (unsigned)A.H && gc0 < A.W) {
      const T* wrow = wplane + (size_t)gr * A.W;
#pragma unroll
      for (int pc = 0; pc < S; ++pc) wreg[pc] = wrow[(unsigned)(gc0 + pc)];
    }
    // left halo columns of 2*lambda*w*r: waves 4 / 5 (columns -1 / -2), one row per lane
    const bool col_task = want_reg && RU > 0 && (wv == 4 || wv == 5) && (wv - 4) < RU;
    T wcol = T(1);
    const int crow = r0 + lane;  // lane < SR: row of this step (virtual step: only its last RU rows)
    const bool col_lane = col_task && lane < SR && (!virt || lane >= SR - RU);
    if (col_lane && wplane != nullptr) {
      const int hgc = C0 - (wv == 4 ? 1 : 2);
      if ((unsigned)crow < (unsigned)A.H && hgc >= 0) wcol = wplane[(size_t)crow * A.W + hgc];
    }
#pragma unroll
    for (int pc = 0; pc < S; ++pc) { acc[pc] = T(0); zown[pc] = T(0); }

    // ---- phase 1 ----
    if (p1_on && (do_z || do_r)) {
      const GT* xb[WIN + 1];
#pragma unroll
      for (int i = 0; i <= WIN; ++i) xb[i] = xs + mwrap<C::NXR>(sx + i) * C::XRG + lane;
      GT* zdst = zs + sz * C::ZRG + lane;
      GT* cdst = cs + sc * C::CRG + lane;
      const bool cost_row = gr >= A.cr0 && gr < A.cr1;
      if (slow)
        m_phase1<T, S, B, REGK, R, true>(A, xb, zdst, cdst, do_z, do_r, count_z, full, cost_row, gr, zrow, CJ0, lane, ybase,
                                          ypre, wreg, acc, zown, cost_data, cost_reg);
      else
        m_phase1<T, S, B, REGK, R, false>(A, xb, zdst, cdst, do_z, do_r, count_z, full, cost_row, gr, zrow, CJ0, lane, ybase,
                                           ypre, wreg, acc, zown, cost_data, cost_reg);
    }
    if (col_lane) {
      // per-lane slots of rows crow .. crow + WIN and of the 2*lambda*w*r row
      int xe[WIN + 1];
      const int q0 = SR * n + lane + C::XLO;  // >= 0 for the lanes of the virtual step that take part
#pragma unroll
      for (int i = 0; i <= WIN; ++i) xe[i] = ((q0 + i + C::NXR) % C::NXR) * (C::XRG * G);
      const int ce = ((SR * n + lane + RU + C::NCR) % C::NCR) * (C::CRG * G);
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
      if (A.g != nullptr) m_phase2<T, S, B, REGK, R>(A, xb2, cb2, zb2, want_data, want_reg, zown, acc);
      if (A.g != nullptr && gr < br1 && gc0 < A.W) {
        T* dst = A.g + (size_t)ch * N + (size_t)gr * A.W + gc0;
#pragma unroll
        for (int pc = 0; pc < S; ++pc) __builtin_nontemporal_store(acc[pc], &dst[pc]);
      }
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
    if (tid == 0) {
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
  if (geo.rr0 != 0 || geo.rr1 != geo.H) return false;
  if (z.n_ring > 0 && (z.ring.rg[0] != 0 || z.ring.rg[1] != 0)) return false;
  if (p->ctx->num_cus <= 0) return false;
  (void)terms;
  const int ns = (geo.w + 63) / 64;
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
