// zhelp_kernel.hpp -- k_eval_zh: the tile kernel of kernels_ztile.hip with HELPER WAVES (included by kernels_ztile.hip, inside
// its anonymous namespace; same arguments, same tiles, same device helpers -- ztile_dev.hpp -- same results).
//
// k_eval_z gives every one of its 8 waves a tile row AND spreads the tile's halo work over them: waves 0 / 1 evaluate the two
// halo rows of zh, waves 2 / 3 the two halo rows of 2*lambda*w*r, waves 4 / 5 the two halo columns, waves 0 .. 4 a second row
// of the x window.  Phase 1 ends when the slowest of them arrives (profiles/r05_phase_clock.txt: 5.7 - 6.8 K cycles against
// 5.2 K for a wave without halo work), and the registers of the halo inputs (a second set of observations, a second set of
// weights, a second window row) are part of every wave's allocation: 124 VGPRs = 4 waves per SIMD.
// Here a workgroup is 8 ROW waves + 2 HELPER waves (640 threads):
//   row wave w      window row w;  own observations, own weights;  phase 1: data term + regulariser pass 1 of tile row w;
//                   phase 2 as in k_eval_z
//   helper 0        window rows 8, 9, 10;  observations of the two zh halo rows;  phase 1: those two rows of zh
//   helper 1        window rows 11, 12 + the halo cells of every window row;  weights of the two 2*lambda*w*r halo rows and
//                   of the halo columns;  phase 1: those two rows, then both halo columns (lanes 0-15 / 16-31 = rows)
// Same LDS (no parked halo weights: 69 KB), two workgroups per CU = 20 waves = 5 per SIMD IF the allocation stays within 96 VGPRs.
#pragma once
#ifndef SRMAP_ZH_DBG
#define SRMAP_ZH_DBG 0   // register probes: 1 = row waves + helper 1 only ... (measurement only)
#endif

namespace zh {
constexpr int kNH = 2;
}

template <typename T, int S, int B, int REGK, int R, bool WD>
__global__ __launch_bounds__(64 * (ZCfg<T, S, B, REGK, R>::TH + zh::kNH), SRMAP_EXP_HELPER_WPE) void k_eval_zh(
    ZArgs<T, B, ZCfg<T, S, B, REGK, R>::NP> A) {
  using C = ZCfg<T, S, B, REGK, R>;
  constexpr int HB = C::HB, NV = C::NV, RU = C::RU, TH = C::TH, NH = zh::kNH, NWV = TH + NH, NT = 64 * NWV;
  constexpr int kBorderLds = (int)((16 * sizeof(int2) + kBorderTabEntries * sizeof(ZEntry) + 16 * sizeof(double) + sizeof(T) - 1) / sizeof(T));
  __shared__ T xs[C::XS_ELEMS > kBorderLds ? C::XS_ELEMS : kBorderLds];
  __shared__ T zs[C::ZS_ELEMS > 0 ? C::ZS_ELEMS : 1];
  __shared__ T cs[C::CS_ELEMS > 0 ? C::CS_ELEMS : 1];
  __shared__ double red[2][NWV];
  static_assert(RU <= 2 && TH + RU <= 16, "halo columns: lanes 0-15 / 16-31 of helper 1 are the rows of column -1 / -2");
  {
    const T* a_x = A.x; const T* a_y = A.y; const T* a_w = A.w; T* a_g = A.g;
    const int a_W = A.W, a_H = A.H, a_wl = A.wl, a_hl = A.hl, a_nby = A.nby, a_E = A.E, a_terms = A.terms, a_obsC = A.obs_C;
    const int a_cr0 = A.cr0, a_cr1 = A.cr1, a_rr0 = A.rr0, a_rr1 = A.rr1, a_sm = A.sel_mode, a_s0 = A.sel0, a_s1 = A.sel1;
    const unsigned a_gx = gridDim.x, a_gy = gridDim.y;
    asm volatile("" ::"s"(a_x), "s"(a_y), "s"(a_w), "s"(a_g), "s"(a_W), "s"(a_H), "s"(a_wl), "s"(a_hl), "s"(a_nby), "s"(a_E),
                 "s"(a_terms), "s"(a_obsC), "s"(a_cr0), "s"(a_cr1), "s"(a_rr0), "s"(a_rr1), "s"(a_sm), "s"(a_s0), "s"(a_s1),
                 "s"(a_gx), "s"(a_gy));
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)blockIdx.y < A.nby) {  // border blocks (NT threads each here)
    if (A.sel_mode == 1) return;
    const int bidx = blockIdx.y * gridDim.x + blockIdx.x;
    const BorderArgs<T>& Bd = *A.bd;
    if (bidx * NT < Bd.n_ring) {
      if (WD && A.fold_xk != nullptr) border_block<T, S, B, NT, WD, WD>(A, Bd, bidx, blockIdx.z, xs, A.nby * gridDim.x);
      else border_block<T, S, B, NT, WD, false>(A, Bd, bidx, blockIdx.z, xs, A.nby * gridDim.x);
    } else if (threadIdx.x == 0) {
      const int nbb = A.nby * gridDim.x;
      put_partial<WD>(A, (size_t)A.n_tile_partials + (size_t)blockIdx.z * nbb + bidx, 0.0, 0.0);
    }
    return;
  }
  const int by = blockIdx.y - A.nby, nby_t = gridDim.y - A.nby;
  int tby = by, tbx = blockIdx.x;
  {  // tile order: as k_eval_z
    const int q = gridDim.x >> 3, rem = gridDim.x & 7;
    const int n0 = blockIdx.x, nbot = (rem == 0) ? (int)gridDim.x - 1 : 8 * q - 1;
    const int n = (nbot > 1) ? (n0 == 1 ? nbot : (n0 == nbot ? 1 : n0)) : n0;
    const int bnd = n & 7;
    tby = bnd * q + (bnd < rem ? bnd : rem) + (n >> 3);
    tbx = (by == 0) ? 0 : (by == 1 ? nby_t - 1 : by - 1);
  }
  if (A.sel_mode != 0 && ((A.sel_mode == 1) != (tby >= A.sel0 && tby < A.sel1))) return;
  const int R0 = tby * TH, CJ0 = tbx * C::CW, C0 = CJ0 * S;
  const int ch = blockIdx.z;
  const size_t N = (size_t)A.W * A.H;
  const size_t nl = (size_t)A.wl * A.hl;
  const bool fold = WD && A.fold_xk != nullptr;
  const T* xplane = (fold ? A.fold_xk : A.x) + (size_t)ch * N;
  const bool helper = wv >= TH;     // uniform
  const int hw = wv - TH;           // helper number
  const int gr = R0 + wv;           // row waves: global HR row of this thread
  const int gc0 = C0 + S * lane;
  const bool want_data = (A.terms & SRMAP_TERM_DATA) != 0;
  const bool want_reg = REGK != 0 && (A.terms & SRMAP_TERM_REG) != 0 && R0 >= A.rr0 && R0 < A.rr1;
  const T* ybase = A.y + (size_t)ch * nl;
  const T* wplane = (want_reg && A.w) ? A.w + (size_t)ch * N : nullptr;
  const DirScale dsc = dir_scale(WD ? A.fold_norms : nullptr);
  const int rm = A.E + HB + 1, cm = (A.E + HB + S) / S + 1;
  const bool edge = (R0 - rm < 0) || (R0 + TH + rm > A.H) || (CJ0 - cm < 0) || (CJ0 + C::CW + cm > A.wl) ||
                    A.cr0 > 0 || A.cr1 < A.H;
  const bool z_halo_on = want_data && B > 1 && A.g != nullptr;
  const bool reg_halo_on = want_reg && A.g != nullptr && RU > 0;
  constexpr int EXTRA = C::XC - C::CW;                 // halo cells of a window row
  constexpr int HROWS = C::XR - TH;                     // window rows the helpers stage
  constexpr int ARH = (HROWS + NH - 1) / NH;            // ... each
  static_assert(TH + NH * ARH > C::XR && C::XR * EXTRA <= 64, "the last helper's last round stages the halo cells");

  // One window row: requests (x, and the direction when the trial point is folded), then -- stage() -- scale / combine and
  // the LDS store.  row: window row; cells: false = the row's CW cells from cell lane, true = the halo cells of ALL rows
  // (lane -> (row, cell)).
  struct RowReq { T v[S]; T vd[S]; T m; bool own; size_t xo; int lrow, lcell; bool live; };
  auto request = [&](int row, bool cells, RowReq& q) __attribute__((always_inline)) {
    const int wrow = cells ? lane / EXTRA : row;
    const int grr = R0 - C::HU + wrow;
    const bool row_in = wrow < C::XR && (unsigned)grr < (unsigned)A.H;
    const int gca = cells ? CJ0 - C::XCL + C::CW + lane % EXTRA : CJ0 - C::XCL + lane;
    const bool ina = row_in && (unsigned)gca < (unsigned)A.wl;
    const T* sa = xplane + (ina ? (size_t)grr * A.W + (size_t)gca * S : (size_t)0);
#pragma unroll
    for (int pc = 0; pc < S; ++pc) q.v[pc] = sa[pc];
#pragma unroll
    for (int pc = 0; pc < S; ++pc) q.vd[pc] = T(0);
    q.own = false; q.xo = 0;
    if (WD && fold) {
      const T* da = A.dvec + (size_t)ch * N + (sa - xplane);
#pragma unroll
      for (int pc = 0; pc < S; ++pc) q.vd[pc] = da[pc];
      const int orow = wrow - C::HU;
      q.own = ina && orow >= 0 && orow < TH && gca >= CJ0 && gca < CJ0 + C::CW;
      q.xo = (size_t)grr * A.W + (size_t)gca * S;
    }
    q.m = ina ? Pre<T>::up(T(1)) : T(0);
    q.lrow = wrow; q.lcell = cells ? C::CW + lane % EXTRA : lane; q.live = wrow < C::XR;
  };
  auto stage = [&](RowReq& q) __attribute__((always_inline)) {
    if (WD && fold) {
      T* xout = A.fold_x + (size_t)ch * N;
#pragma unroll
      for (int pc = 0; pc < S; ++pc) q.v[pc] = q.v[pc] + A.fold_stp * dir_elem<T>(q.vd[pc], dsc);
      if (q.own) {
#pragma unroll
        for (int pc = 0; pc < S; ++pc) xout[q.xo + pc] = q.v[pc];
      }
    }
    if (q.live) {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) xs[q.lrow * C::XROW + pc * C::XC + q.lcell] = q.v[pc] * q.m;
    }
  };

  T acc[S], zown[S];
#pragma unroll
  for (int j = 0; j < S; ++j) { acc[j] = T(0); zown[j] = T(0); }
  double cost_data = 0.0, cost_reg = 0.0;
  T mk[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) mk[pc] = T(1);
  const bool reg_border = (R0 + TH + C::WIN > A.H) || (C0 + C::TW + C::WIN > A.W);

  if (!helper && SRMAP_ZH_DBG != 3 && SRMAP_ZH_DBG != 2) {
    // =============================== ROW WAVES ===============================
    RowReq q0;
    request(wv, false, q0);
    T ypre[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) ypre[v] = T(0);
    if (want_data) z_row_prefetch<T, S, B, C>(A, wv, R0, CJ0, lane, edge, ybase, ypre);
    T wreg[S];
#pragma unroll
    for (int pc = 0; pc < S; ++pc) wreg[pc] = T(1);
    if (wplane != nullptr && gr < A.H && gc0 < A.W) {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) wreg[pc] = wplane[(size_t)gr * A.W + gc0 + pc];
    }
    stage(q0);
    __syncthreads();
#pragma unroll
    for (int pc = 0; pc < S; ++pc) mk[pc] = (gr < A.H && gc0 + pc < A.W) ? T(1) : T(0);
    if (want_data) {
      if (edge) z_row<T, S, B, C, true>(A, xs, zs, wv, R0, CJ0, lane, ybase, true, ypre, true, mk, zown, cost_data);
      else z_row<T, S, B, C, false>(A, xs, zs, wv, R0, CJ0, lane, ybase, true, ypre, true, mk, zown, cost_data);
    }
    if (want_reg) {
      const bool cost_row = gr >= A.cr0 && gr < A.cr1;
      if (reg_border)
        reg_row<T, S, REGK, R, C, true, true>(acc, cost_reg, xs, cs, wreg, wv, lane, gr, gc0, A.W, A.H, A.lambda, A.powtab, A.pwsum, cost_row);
      else
        reg_row<T, S, REGK, R, C, false, true>(acc, cost_reg, xs, cs, wreg, wv, lane, gr, gc0, A.W, A.H, A.lambda, A.powtab, A.pwsum, cost_row);
    }
  } else if (SRMAP_ZH_DBG != 1 && SRMAP_ZH_DBG != 4 && hw == 0) {
    // =============================== HELPER 0: window rows TH .. TH + ARH - 1, the two halo rows of zh ===============================
    RowReq q[ARH];
#pragma unroll
    for (int it = 0; it < ARH; ++it) request(TH + it, false, q[it]);
    T ypa[NV], ypb[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { ypa[v] = T(0); ypb[v] = T(0); }
    if (z_halo_on) {
      z_row_prefetch<T, S, B, C>(A, -HB, R0, CJ0, lane, edge, ybase, ypa);
      z_row_prefetch<T, S, B, C>(A, TH - 1 + HB, R0, CJ0, lane, edge, ybase, ypb);
    }
#pragma unroll
    for (int it = 0; it < ARH; ++it) stage(q[it]);
    __syncthreads();
    if (z_halo_on) {
      T dummy[S];
      double dcost = 0.0;
      if (edge) {
        z_row<T, S, B, C, true>(A, xs, zs, -HB, R0, CJ0, lane, ybase, true, ypa, false, mk, dummy, dcost);
        z_row<T, S, B, C, true>(A, xs, zs, TH - 1 + HB, R0, CJ0, lane, ybase, true, ypb, false, mk, dummy, dcost);
      } else {
        z_row<T, S, B, C, false>(A, xs, zs, -HB, R0, CJ0, lane, ybase, true, ypa, false, mk, dummy, dcost);
        z_row<T, S, B, C, false>(A, xs, zs, TH - 1 + HB, R0, CJ0, lane, ybase, true, ypb, false, mk, dummy, dcost);
      }
    }
  } else if (SRMAP_ZH_DBG == 0 || SRMAP_ZH_DBG == 3 || SRMAP_ZH_DBG == 1) {
    // =============================== HELPER 1: the remaining window rows + the halo cells of every row; the halo rows and
    // columns of 2*lambda*w*r ===============================
    RowReq q[ARH];
#pragma unroll
    for (int it = 0; it < ARH; ++it) {
      const int row = TH + ARH + it;
      if (it == ARH - 1) request(0, true, q[it]);   // the halo cells of all rows (this slot has no row of its own)
      else request(row, false, q[it]);
    }
    T wha[S], whb[S];
#pragma unroll
    for (int pc = 0; pc < S; ++pc) { wha[pc] = T(1); whb[pc] = T(1); }
    T wcol = T(1);
    if (reg_halo_on && wplane != nullptr) {
      if (R0 - 1 >= 0 && gc0 < A.W) {
#pragma unroll
        for (int pc = 0; pc < S; ++pc) wha[pc] = wplane[(size_t)(R0 - 1) * A.W + gc0 + pc];
      }
      if (RU >= 2 && R0 - 2 >= 0 && gc0 < A.W) {
#pragma unroll
        for (int pc = 0; pc < S; ++pc) whb[pc] = wplane[(size_t)(R0 - 2) * A.W + gc0 + pc];
      }
      if (lane < 16 * RU && (lane & 15) < TH + RU) {
        const int hgr = R0 + (lane & 15) - RU, hgc = C0 - 1 - (lane >> 4);
        if (hgr >= 0 && hgr < A.H && hgc >= 0) wcol = wplane[(size_t)hgr * A.W + hgc];
      }
    }
#pragma unroll
    for (int it = 0; it < ARH; ++it) stage(q[it]);
    __syncthreads();
    if (reg_halo_on) {
      T dacc[S];
      double dc = 0.0;
      const bool hb_border = C0 + C::TW + C::WIN > A.W || R0 + C::WIN > A.H;
      if (hb_border) {
        reg_row<T, S, REGK, R, C, true, false>(dacc, dc, xs, cs, wha, -1, lane, R0 - 1, gc0, A.W, A.H, A.lambda, A.powtab, A.pwsum, false);
        if (RU >= 2) reg_row<T, S, REGK, R, C, true, false>(dacc, dc, xs, cs, whb, -2, lane, R0 - 2, gc0, A.W, A.H, A.lambda, A.powtab, A.pwsum, false);
      } else {
        reg_row<T, S, REGK, R, C, false, false>(dacc, dc, xs, cs, wha, -1, lane, R0 - 1, gc0, A.W, A.H, A.lambda, A.powtab, A.pwsum, false);
        if (RU >= 2) reg_row<T, S, REGK, R, C, false, false>(dacc, dc, xs, cs, whb, -2, lane, R0 - 2, gc0, A.W, A.H, A.lambda, A.powtab, A.pwsum, false);
      }
      // halo columns: lanes 0-15 = the rows of column -1, lanes 16-31 those of column -2
      const int rowrel = (lane & 15) - RU;
      const int lo = rowrel * C::XROW, lc = rowrel * C::CROW;
      const bool rows_ok = (lane & 15) < TH + RU;
      if (lane < 16 && rows_ok) {
        if (reg_border) reg_halo_col<T, S, REGK, R, C, -1, true>(xs + lo, cs + lc, wcol, rowrel, R0, C0, A.W, A.H, A.lambda, A.powtab);
        else reg_halo_col<T, S, REGK, R, C, -1, false>(xs + lo, cs + lc, wcol, rowrel, R0, C0, A.W, A.H, A.lambda, A.powtab);
      }
      if (RU >= 2 && lane >= 16 && lane < 32 && rows_ok) {
        if (reg_border) reg_halo_col<T, S, REGK, R, C, -2, true>(xs + lo, cs + lc, wcol, rowrel, R0, C0, A.W, A.H, A.lambda, A.powtab);
        else reg_halo_col<T, S, REGK, R, C, -2, false>(xs + lo, cs + lc, wcol, rowrel, R0, C0, A.W, A.H, A.lambda, A.powtab);
      }
    }
  }
  __syncthreads();

  // ---------------- phase 2 (row waves) ----------------
  double gd = 0.0;
  if (!helper) {
    T dreg[S];
#pragma unroll
    for (int pc = 0; pc < S; ++pc) dreg[pc] = T(0);
    if (WD && gr < A.H && gc0 < A.W && gr >= A.cr0 && gr < A.cr1) {
      const T* dp = A.dvec + (size_t)ch * N + (size_t)gr * A.W + gc0;
#pragma unroll
      for (int pc = 0; pc < S; ++pc) dreg[pc] = dir_elem<T>(dp[pc], dsc);
    }
    if (want_data && A.g != nullptr) {
      const T sc = (T)(2 * S * S);
#pragma unroll
      for (int pc = 0; pc < S; ++pc) {
        T zz;
        if (B == 1) {
          zz = zown[pc];
        } else {
          zz = T(0);
#pragma unroll
          for (int a = 0; a < B; ++a) zz += k1_tap<B>(A, a) * zs[(wv + a) * C::ZROW + pc * C::CW + lane];
        }
        acc[pc] += sc * zz;
      }
    }
    if (want_reg && A.g != nullptr) reg_pass2z<T, S, REGK, R, C>(acc, xs, cs, wv, lane, A.powtab);
    if (A.g != nullptr && gr < A.H && gc0 < A.W) {
      T* dst = A.g + (size_t)ch * N + (size_t)gr * A.W + gc0;
#pragma unroll
      for (int pc = 0; pc < S; ++pc) __builtin_nontemporal_store(acc[pc], &dst[pc]);
    }
    if (WD) {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) gd += (double)acc[pc] * (double)dreg[pc];
    }
  }
  // ---------------- cost partial of this workgroup ----------------
  {
    if (WD) gd = wave_sum_d(gd);
    const double cw = wave_sum_d((double)(S * S) * cost_data + cost_reg);
    if (lane == 0) { red[0][wv] = cw; if (WD) red[1][wv] = gd; }
    __syncthreads();
    if (tid == 0) {
      double c = 0.0, d = 0.0;
#pragma unroll
      for (int i = 0; i < TH; ++i) { c += red[0][i]; if (WD) d += red[1][i]; }
      const size_t b = ((size_t)blockIdx.z * nby_t + by) * gridDim.x + blockIdx.x;
      put_partial<WD>(A, b, c, d);
    }
    if (A.mfinish && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && blockIdx.z == gridDim.z - 1) {
      __syncthreads();
      finish_block<WD, NT>(A, &red[0][0]);
    }
  }
}
