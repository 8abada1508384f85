// kernels_ztile.hip -- the hot path: one MAP gradient iteration
// (ObjectiveFunction::ComputeAllTerms, objective_function.cpp:5-20) as ONE
// LDS-tiled launch (border blocks ride at the front of its grid; the cost
// partials are reduced by its last workgroup -- a finish launch follows only when
// in-image border corrections have to be subtracted), for the common geometry:
// integer motion shifts, HR = LR * S, S in {2,3,4}, blur size B in {1,3}, first
// regulariser 2-D TV or BTV with range <= 3.  Everything else is evaluated by
// kernels_direct.hip.
//
// Formulation.  With integer shifts the warp M_k and the
// blur B are shift-invariant, so AWAY FROM THE IMAGE BORDER they commute and the
// data term (objective_data_term.cpp:15-116, image_model.cpp:86-101) collapses
// onto the HR grid:
//     r_k(i,j) = (B x)(p) - y_k(i,j),   p = (S i + oy_k, S j + ox_k)   ("z position")
//     z(p)     = sum over the frames k whose LR grid hits p of r_k
//     g_data   = 2 S^2 * B^T z,         cost_data = S^2 * sum r_k^2
// i.e. every residual is evaluated ONCE, by the thread that owns HR pixel p
// ("owner computes"), from a full-resolution blur of x at its own pixels; the
// K per-frame transposes become one blur of z.  Per HR pixel this is 9 + 9 FMAs
// and K/S^2 observation loads instead of K gathers.  The reference clips every
// stage to the H x W domain separately (SURVEY.md section 8a'); in owner-computes
// form the clips are:
//   (a) warp zero fill: x outside the image reads 0            -> zero-filled tile;
//   (b) blur zero padding on the WARPED image: blur taps whose warped coordinate
//       leaves the image are dropped.  With B <= S + 1 that is blur tap row 0 of LR
//       row 0 and tap column 0 of LR column 0 only              -> subtracted for those
//       residuals in tiles at the top / left image edge (EDGE code path);
//   (c) LR pixels exist only inside the LR image                -> validity mask (EDGE);
//   (d) the transpose warp clips its SOURCE: frame k contributes to output pixel q
//       only if q - o_k is inside the image.  That depends on (q, k), so it cannot
//       be folded into the frame-summed z: the tiles add every frame and the
//       border blocks collect the excluded contributions for the pixels within
//       E = max|shift| of the image edge (k_finish_eval subtracts them), and the
//       cost of the residuals whose z position lies outside the image (they have
//       no owner).
//
// Tile kernel k_eval_z: a workgroup of 8 waves owns 8 HR rows x 64*S columns;
// wave = HR row, lane = LR cell, a thread owns the S consecutive pixels of its
// cell.  x tile (+halo) in LDS in polyphase layout xs[row][col mod S][cell]
// (unit stride across lanes for every window read, offsets are immediates).
//   phase 1  B x at S+2 pixels (own + one neighbour each side), residuals of
//            the matching frames (host-built table per (row phase, col phase)),
//            horizontal half of B^T in registers -> zh to LDS; regulariser
//            pass 1 (values, self term, 2*lambda*w*r -> LDS); halo rows of zh /
//            2*lambda*w*r by whole waves (0-1 / 2-3), the left halo columns one per
//            wave (4 / 5, lanes = rows); their IRLS weights wait in LDS;
//   phase 2  vertical half of B^T from zh; regulariser pass 2; g store.
// The frame table's counts and round 0 travel by value in the kernel arguments
// (scalar loads only); later rounds are read through the constant address space.
// No MFMA: stencil path.  Cost partials are reduced in fixed order
// (deterministic).


#include "ztile_dev.hpp"
#ifndef SRMAP_EXP_XT
#define SRMAP_EXP_XT 1
#endif

// Build-time switch of the measurement builds (tools/exp_build.sh); the product build does not define it.
//   SRMAP_ZT_ONLY_CFG2  instantiate only k_eval_z<double, 4, 3, BTV, 3> (seconds instead of minutes per variant)
//   SRMAP_EXP_NOLOAD    TIMING ONLY (results wrong by construction): every global load of a tile workgroup replaced by a
//                       value formed in registers -- the time no prefetch scheme can beat (profiles/r05_ceiling.txt)
//   SRMAP_EXP_NOHALO    TIMING ONLY: no halo-row / halo-column passes (what a marching band saves at best)
//   SRMAP_ZT_ONLY_T     with SRMAP_ZT_ONLY_CFG2: the arithmetic type of that one instance (default double)
//   SRMAP_EXP_F32_WPE   f32 instances: waves per SIMD the register budget is set for (product: 6 = 80 VGPRs)
#ifndef SRMAP_ZT_ONLY_T
#define SRMAP_ZT_ONLY_T double
#endif
//   SRMAP_ZT_ONLY_B     with SRMAP_ZT_ONLY_CFG2: the blur size of that one instance (default 3; 1 = the cfg3 / cfg5 instance)
#ifndef SRMAP_ZT_ONLY_B
#define SRMAP_ZT_ONLY_B 3
#endif
#ifndef SRMAP_EXP_F32_WPE
#define SRMAP_EXP_F32_WPE 6
#endif
//   SRMAP_EXP_SPOLD     sub-pixel instances: the entry-major tap table of rounds 2-4 (one request per pixel and tap)
//                       instead of the source-major one (z_row_sp2)
#ifndef SRMAP_EXP_SPOLD
#define SRMAP_EXP_SPOLD 0
#endif
//   SRMAP_EXP_F64_WPE   f64 S > 2 instances: waves per SIMD of __launch_bounds__ (product: 4)
//   SRMAP_EXP_ALIAS_ZC  TIMING ONLY: zh and 2*lambda*w*r share one LDS array (what a three-phase tile would allocate)
#ifndef SRMAP_EXP_F64_WPE
#define SRMAP_EXP_F64_WPE 4
#endif
#ifndef SRMAP_EXP_ALIAS_ZC
#define SRMAP_EXP_ALIAS_ZC 0
#endif
//   SRMAP_EXP_LDS_PAD   TIMING ONLY: extra bytes of (dynamic) LDS per workgroup (fewer resident workgroups, same instruction stream)
#ifndef SRMAP_EXP_LDS_PAD
#define SRMAP_EXP_LDS_PAD 0
#endif
//   SRMAP_EXP_REGFIRST  phase 1: regulariser pass before the data term (the observations get a pass longer to arrive)
//   SRMAP_EXP_NTLOAD    observations and IRLS weights requested with non-temporal loads (streamed once)
//   SRMAP_EXP_HEADPRIO  s_setprio 3 from the start of a tile workgroup until its requests are issued
#ifndef SRMAP_EXP_REGFIRST
#define SRMAP_EXP_REGFIRST 0
#endif
#ifndef SRMAP_EXP_NTLOAD
#define SRMAP_EXP_NTLOAD 0
#endif
#ifndef SRMAP_EXP_HEADPRIO
#define SRMAP_EXP_HEADPRIO 0
#endif
#ifndef SRMAP_EXP_NOLOAD
#define SRMAP_EXP_NOLOAD 0
#endif
#ifndef SRMAP_EXP_NOHALO
#define SRMAP_EXP_NOHALO 0
#endif
// The TIMING-ONLY switches produce wrong results by construction: a build that sets one has to say that it is a
// measurement build (tools/exp_build.sh, tools/full_build.sh and tools/phase_clock/build.sh define it; the product build does not).
#if (SRMAP_EXP_NOLOAD || SRMAP_EXP_NOHALO || SRMAP_EXP_ALIAS_ZC || SRMAP_EXP_LDS_PAD || defined(SRMAP_EXP_SPNOLOAD)) && !defined(SRMAP_MEASUREMENT_BUILD)
#error "SRMAP_EXP_NOLOAD / NOHALO / ALIAS_ZC / LDS_PAD / SPNOLOAD are timing-only switches: define SRMAP_MEASUREMENT_BUILD (tools/exp_build.sh)"
#endif

namespace srmap {

namespace {

template <typename T, int S, int B, int REGK, int R, bool WD, bool SP>
__global__ __launch_bounds__((ZCfg<T, S, B, REGK, R>::NT), (sizeof(T) == 4 ? SRMAP_EXP_F32_WPE : (S == 2 ? 6 : SRMAP_EXP_F64_WPE))) void k_eval_z(
    ZArgs<T, B, ZCfg<T, S, B, REGK, R>::NP> A) {
  using C = ZCfg<T, S, B, REGK, R>;
  constexpr int HB = C::HB, NV = C::NV, RU = C::RU;
  // border blocks borrow the x tile's LDS for the frame table
  constexpr int kBorderLds = (int)((16 * sizeof(int2) + kBorderTabEntries * sizeof(ZEntry) + 16 * sizeof(double) + sizeof(T) - 1) / sizeof(T));
  __shared__ T xs[C::XS_ELEMS > kBorderLds ? C::XS_ELEMS : kBorderLds];
#if SRMAP_EXP_ALIAS_ZC
  __shared__ T zcs[C::ZS_ELEMS > C::CS_ELEMS ? C::ZS_ELEMS : (C::CS_ELEMS > 0 ? C::CS_ELEMS : 1)];
  T* const zs = zcs;
  T* const cs = zcs;
#else
  __shared__ T zs[C::ZS_ELEMS > 0 ? C::ZS_ELEMS : 1];
  __shared__ T cs[C::CS_ELEMS > 0 ? C::CS_ELEMS : 1];
#endif
  __shared__ double red[2][C::NW];
  __shared__ T wcs[32];  // IRLS weights of the left-halo-column pixels (two columns x up to 16 rows)
  __shared__ T whs[(C::RU > 0 ? C::RU : 1) * S * C::CW];  // IRLS weights of the halo rows of 2*lambda*w*r

  // Every argument the head of a workgroup reads -- up to its first memory requests -- is requested HERE, in one batch
  // ahead of the first branch.  Left to the compiler each uniform early-out below (border block? selected tile row?
  // edge tile?) fetched its own word behind its own s_waitcnt: six dependent scalar-load round trips (~250 cycles
  // each) stood between the start of a workgroup and its first x request (profiles/r03_phase_clock.txt: "issue x
  // loads" 2.1 K cycles).  The empty asm makes all of them live at this point; later reads of the same fields reuse
  // the registers.
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
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave = HR row of the tile (SGPR)
  // XCD-aware tile order: workgroups are dealt to the 8 XCDs round robin in launch order and every XCD has its
  // own L2; with the tile ROW on blockIdx.x and launch index n -> row band (n mod 8) the workgroups an XCD runs at
  // the same time are vertical neighbours and share their x halo rows in that L2.  Bijective for any row count.
  if ((int)blockIdx.y < A.nby) {  // border blocks come first in dispatch order (uniform branch)
    if (A.sel_mode == 1) return;
    const int bidx = blockIdx.y * gridDim.x + blockIdx.x;
    const BorderArgs<T>& Bd = *A.bd;
    if (bidx * C::NT < Bd.n_ring) {
      if (WD && A.fold_xk != nullptr) border_block<T, S, B, C::NT, WD, WD>(A, Bd, bidx, blockIdx.z, xs, A.nby * gridDim.x);
      else border_block<T, S, B, C::NT, WD, false>(A, Bd, bidx, blockIdx.z, xs, A.nby * gridDim.x);
    }
    else if (threadIdx.x == 0) {
      const int nbb = A.nby * gridDim.x;
      put_partial<WD>(A, (size_t)A.n_tile_partials + (size_t)blockIdx.z * nbb + bidx, 0.0, 0.0);
    }
    return;
  }
  const int by = blockIdx.y - A.nby, nby_t = gridDim.y - A.nby;
  int tby = by, tbx = blockIdx.x;
  {
    // ... with launch index 1 and the launch index of the BOTTOM tile row swapped: the bottom row (masked edge path in
    // every column, the slowest tiles) would otherwise be among the last workgroups of every tile column -- of the last
    // column too, where it sets the end of the launch.  It is dispatched second now, like the top row first.  The
    // bottom row is the last row of band 7: launch index 8 q - 1 (= the last index when the row count is a multiple
    // of 8; with a remainder the last index maps to another band's last row).
    const int q = gridDim.x >> 3, rem = gridDim.x & 7;
    const int n0 = blockIdx.x, nbot = (rem == 0) ? (int)gridDim.x - 1 : 8 * q - 1;
    const int n = (nbot > 1) ? (n0 == 1 ? nbot : (n0 == nbot ? 1 : n0)) : n0;
    const int bnd = n & 7;
    tby = bnd * q + (bnd < rem ? bnd : rem) + (n >> 3);
    // tile columns in the order first, last, second, ...: the masked edge columns (longest-lived tiles) are not the
    // launch's last generation
    tbx = (by == 0) ? 0 : (by == 1 ? nby_t - 1 : by - 1);
  }
  if (A.sel_mode != 0 && ((A.sel_mode == 1) != (tby >= A.sel0 && tby < A.sel1))) return;  // uniform; before any barrier
  if (SRMAP_EXP_HEADPRIO) __builtin_amdgcn_s_setprio(3);
  const int R0 = tby * C::TH, CJ0 = tbx * C::CW, C0 = CJ0 * S;
  const int ch = blockIdx.z;
  const size_t N = (size_t)A.W * A.H;
  const size_t nl = (size_t)A.wl * A.hl;
  const bool fold = WD && A.fold_xk != nullptr;   // uniform: the window is xk + stp * d (solver line search)
  const T* xplane = (fold ? A.fold_xk : A.x) + (size_t)ch * N;
  const int gr = R0 + wv;          // global HR row of this thread
  const int gc0 = C0 + S * lane;   // first global HR column of this thread
  const int cellg = CJ0 + lane;
  const bool want_data = (A.terms & SRMAP_TERM_DATA) != 0;
  // frame shards evaluate the regulariser of their own row band only (whole tiles: the band is tile aligned)
  const bool want_reg = REGK != 0 && (A.terms & SRMAP_TERM_REG) != 0 && R0 >= A.rr0 && R0 < A.rr1;
  const T* ybase = A.y + (size_t)ch * nl;
  // ---------------- global loads whose addresses are known now: x tile, observations, IRLS weights ----------------
  constexpr int ARI = (C::XR + C::NW - 1) / C::NW;  // x rows per wave
  constexpr int EXTRA = C::XC - C::CW;              // halo cells, staged by the first lanes
  // XT: the EXTRA halo cells of ALL rows are requested by the last wave in its (otherwise idle) last round -- one lane per
  // (row, cell) -- instead of by the first EXTRA lanes of every wave in every round (half of a wave's x requests
  // served two lanes each)
  constexpr bool XT = SRMAP_EXP_XT != 0 && ARI >= 2 && (C::NW - 1) + (ARI - 1) * C::NW >= C::XR && C::XR * EXTRA <= 64;
  T va[ARI][S], vb[ARI][S], ma[ARI], mb[ARI];
  T vda[WD ? ARI : 1][S], vdb[WD ? ARI : 1][S];   // WD: the direction at the window's elements (fold)
  bool owna[ARI] = {}, ownb[ARI] = {};
  size_t xoa[ARI] = {}, xob[ARI] = {};
#pragma unroll
  for (int it = 0; it < ARI; ++it) {
    const bool xt_slot = XT && it == ARI - 1 && wv == C::NW - 1;  // uniform
    const int row = xt_slot ? lane / EXTRA : wv + it * C::NW;
    const int grr = R0 - C::HU + row;
    const bool row_in = row < C::XR && (unsigned)grr < (unsigned)A.H;  // uniform (XT slot: per lane)
    const int gca = xt_slot ? CJ0 - C::XCL + C::CW + lane % EXTRA : CJ0 - C::XCL + lane, gcb = gca + C::CW;
    const bool ina = row_in && (unsigned)gca < (unsigned)A.wl;
    const bool inb = !XT && row_in && lane < EXTRA && (unsigned)gcb < (unsigned)A.wl;
    const T* sa = xplane + (ina ? (size_t)grr * A.W + (size_t)gca * S : (size_t)0);
    const T* sb = xplane + (inb ? (size_t)grr * A.W + (size_t)gcb * S : (size_t)0);
#pragma unroll
    for (int pc = 0; pc < S; ++pc) va[it][pc] = SRMAP_EXP_NOLOAD ? (T)(lane + pc + it) * A.lambda : sa[pc];
    if (!XT) {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) vb[it][pc] = sb[pc];
    }
    if (WD) {  // the direction at the same elements (requested with the window; combined when the window goes to LDS)
#pragma unroll
      for (int pc = 0; pc < S; ++pc) { vda[it][pc] = T(0); vdb[it][pc] = T(0); }
      if (fold) {
        const T* dpl = A.dvec + (size_t)ch * N;
        const T* da = dpl + (sa - xplane);
        const T* db = dpl + (sb - xplane);
#pragma unroll
        for (int pc = 0; pc < S; ++pc) vda[it][pc] = da[pc];
        if (!XT) {
#pragma unroll
          for (int pc = 0; pc < S; ++pc) vdb[it][pc] = db[pc];
        }
        // where the window element is one of the tile's OWN pixels (not a halo element, inside the image) the trial point
        // is also written out: the solver's x holds it after the evaluation, as after k_axpy_out
        const int orow = row - C::HU;
        owna[it] = ina && orow >= 0 && orow < C::TH && gca >= CJ0 && gca < CJ0 + C::CW;
        ownb[it] = inb && orow >= 0 && orow < C::TH && gcb >= CJ0 && gcb < CJ0 + C::CW;
        xoa[it] = (size_t)grr * A.W + (size_t)gca * S;
        xob[it] = (size_t)grr * A.W + (size_t)gcb * S;
      }
    }
    // scale 2^Q inside the image, 0 outside: applied as ONE multiply when the tile goes to LDS -- a select on the
    // loaded value makes the compiler wait for this row group before it requests the next one
    ma[it] = ina ? Pre<T>::up(T(1)) : T(0);
    mb[it] = inb ? Pre<T>::up(T(1)) : T(0);
  }
  T ypre[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) ypre[v] = T(0);
  // tiles whose residuals can touch LR row / column 0 or leave the LR image take the masked path (uniform);
  // so do row-band problems (cost rows restricted)
  const int rm = A.E + HB + 1, cm = (A.E + HB + S) / S + 1;
  const bool edge = (R0 - rm < 0) || (R0 + C::TH + rm > A.H) || (CJ0 - cm < 0) || (CJ0 + C::CW + cm > A.wl) ||
                    A.cr0 > 0 || A.cr1 < A.H;  // partial tiles (bottom / right) are edge tiles by the first two tests
  const int hrowz = wv == 0 ? -HB : C::TH - 1 + HB;  // halo rows of zh: tile rows -1 (wave 0) and TH (wave 1)
  const bool has_z_halo = !SRMAP_EXP_NOHALO && want_data && B > 1 && A.g != nullptr && wv < 2;
  T ypre2[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) ypre2[v] = T(0);
  if (SRMAP_EXP_NOLOAD) {
#pragma unroll
    for (int v = 0; v < NV; ++v) { ypre[v] = (T)(lane + v) * A.lambda; ypre2[v] = (T)(lane - v) * A.lambda; }
  } else {
    if (!SP && want_data) z_row_prefetch<T, S, B, C>(A, wv, R0, CJ0, lane, edge, ybase, ypre);
    if (!SP && has_z_halo) z_row_prefetch<T, S, B, C>(A, hrowz, R0, CJ0, lane, edge, ybase, ypre2);
  }
  const T* wplane = (want_reg && A.w) ? A.w + (size_t)ch * N : nullptr;
  T wreg[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) wreg[pc] = T(1);
  if (SRMAP_EXP_NOLOAD) {
#pragma unroll
    for (int pc = 0; pc < S; ++pc) wreg[pc] = (T)(lane + 2 * pc + 1) * A.lambda;
  } else if (wplane != nullptr && gr < A.H && gc0 < A.W) {
#pragma unroll
    for (int pc = 0; pc < S; ++pc) wreg[pc] = SRMAP_EXP_NTLOAD ? __builtin_nontemporal_load(&wplane[(size_t)gr * A.W + gc0 + pc]) : wplane[(size_t)gr * A.W + gc0 + pc];
  }
  // halo row of 2*lambda*w*r this wave evaluates (waves 2 .. 2+RU-1: tile rows -1 .. -RU) and its weights
  const bool reg_halo_on = !SRMAP_EXP_NOHALO && want_reg && A.g != nullptr && RU > 0;
  const int hrow = -(wv - 1);  // wave 2 -> -1, wave 3 -> -2
  const bool has_reg_halo = reg_halo_on && wv >= 2 && wv < 2 + RU;
  T whalo[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) whalo[pc] = T(1);
  if (!SRMAP_EXP_NOLOAD && has_reg_halo && wplane != nullptr && R0 + hrow >= 0 && gc0 < A.W) {
#pragma unroll
    for (int pc = 0; pc < S; ++pc) whalo[pc] = wplane[(size_t)(R0 + hrow) * A.W + gc0 + pc];
  }

  // weight of the left-halo-column pixel this lane evaluates in phase 1 (waves 4 / 5, one row per lane): requested
  // with the tile's other inputs and parked in LDS with the x tile -- a load inside the task stalled these two waves,
  // and with them the workgroup's second barrier, for a memory round trip
  T wcolv = T(1);
  const bool col_task = reg_halo_on && (wv == 4 || wv == 5) && lane < C::TH + RU;
  if (!SRMAP_EXP_NOLOAD && col_task && wplane != nullptr) {
    const int hgr = R0 + lane - RU, hgc = C0 - (wv == 4 ? 1 : 2);
    if (hgr >= 0 && hgr < A.H && hgc >= 0) wcolv = wplane[(size_t)hgr * A.W + hgc];
  }

  // WD: how elements of the direction are read (as given, or normalised on the fly from the unnormalised direction)
  const DirScale dsc = dir_scale(WD ? A.fold_norms : nullptr);
  if (WD && fold) {  // trial point: x = xk + stp * d, element by element the expression of solver.hip's k_axpy_out
    T* xout = A.fold_x + (size_t)ch * N;
#pragma unroll
    for (int it = 0; it < ARI; ++it) {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) va[it][pc] = va[it][pc] + A.fold_stp * dir_elem<T>(vda[it][pc], dsc);
      if (owna[it]) {
#pragma unroll
        for (int pc = 0; pc < S; ++pc) xout[xoa[it] + pc] = va[it][pc];
      }
      if (!XT) {
#pragma unroll
        for (int pc = 0; pc < S; ++pc) vb[it][pc] = vb[it][pc] + A.fold_stp * dir_elem<T>(vdb[it][pc], dsc);
        if (ownb[it]) {
#pragma unroll
          for (int pc = 0; pc < S; ++pc) xout[xob[it] + pc] = vb[it][pc];
        }
      }
    }
  }
  if (SRMAP_EXP_HEADPRIO) __builtin_amdgcn_s_setprio(0);
  // ---------------- x tile -> LDS, polyphase ----------------
#pragma unroll
  for (int it = 0; it < ARI; ++it) {
    const bool xt_slot = XT && it == ARI - 1 && wv == C::NW - 1;  // uniform
    if (xt_slot) {
      const int xrow = lane / EXTRA;
      if (xrow < C::XR) {
#pragma unroll
        for (int pc = 0; pc < S; ++pc) xs[xrow * C::XROW + pc * C::XC + C::CW + lane % EXTRA] = va[it][pc] * ma[it];
      }
      continue;
    }
    const int row = wv + it * C::NW;
    if (row < C::XR) {  // uniform
#pragma unroll
      for (int pc = 0; pc < S; ++pc) xs[row * C::XROW + pc * C::XC + lane] = va[it][pc] * ma[it];
      if (!XT && lane < EXTRA) {
#pragma unroll
        for (int pc = 0; pc < S; ++pc) xs[row * C::XROW + pc * C::XC + C::CW + lane] = vb[it][pc] * mb[it];
      }
    }
  }
  if (col_task) wcs[(wv - 4) * 16 + lane] = wcolv;
  if (has_reg_halo) {  // the halo row's weights wait in LDS too (eight registers less across the data term)
#pragma unroll
    for (int pc = 0; pc < S; ++pc) whs[((wv - 2) * S + pc) * C::CW + lane] = whalo[pc];
  }
  __syncthreads();

  // in-image mask of this thread's pixels (partial tiles at the right / bottom edge)
  T mk[S];
#pragma unroll
  for (int pc = 0; pc < S; ++pc) mk[pc] = (gr < A.H && gc0 + pc < A.W) ? T(1) : T(0);

  T acc[S], zown[S];
#pragma unroll
  for (int j = 0; j < S; ++j) { acc[j] = T(0); zown[j] = T(0); }
  double cost_data = 0.0, cost_reg = 0.0;

  // ---------------- phase 1: data term ----------------
  auto phase1_data = [&]() __attribute__((always_inline)) {
  if (SP && want_data) {
    T dummy[S];
    // table rows reach (Dr + S) / S LR rows around the tile's own: only the top / bottom tile rows can leave the image
    const bool row_edge = (R0 - A.Dr - 2 * S < 0) || (R0 + C::TH + A.Dr + 2 * S > A.H);
    // table columns reach Dr / S + 1 LR cells around a pixel's own (+ 1 for the neighbour pixels of the cell)
    const int mj = A.Dr / S + 3;
    const bool col_inner = (CJ0 - mj >= 0) && (CJ0 + C::CW + mj <= A.wl);
#if SRMAP_EXP_SPOLD
#define SRMAP_ZROW_SP z_row_sp
#else
#define SRMAP_ZROW_SP z_row_sp2
#endif
    if (row_edge) {
      SRMAP_ZROW_SP<T, S, B, C, true, true>(A, zs, wv, R0, CJ0, lane, ch, zown);
      if (has_z_halo) SRMAP_ZROW_SP<T, S, B, C, true, true>(A, zs, hrowz, R0, CJ0, lane, ch, dummy);
    } else if (!col_inner) {
      SRMAP_ZROW_SP<T, S, B, C, false, true>(A, zs, wv, R0, CJ0, lane, ch, zown);
      if (has_z_halo) SRMAP_ZROW_SP<T, S, B, C, false, true>(A, zs, hrowz, R0, CJ0, lane, ch, dummy);
    } else {
      SRMAP_ZROW_SP<T, S, B, C, false, false>(A, zs, wv, R0, CJ0, lane, ch, zown);
      if (has_z_halo) SRMAP_ZROW_SP<T, S, B, C, false, false>(A, zs, hrowz, R0, CJ0, lane, ch, dummy);
    }
#undef SRMAP_ZROW_SP
  }
  if (!SP && want_data) {
    T dummy[S];
    double dcost = 0.0;
    if (edge) {
      z_row<T, S, B, C, true>(A, xs, zs, wv, R0, CJ0, lane, ybase, true, ypre, true, mk, zown, cost_data);
      if (has_z_halo) z_row<T, S, B, C, true>(A, xs, zs, hrowz, R0, CJ0, lane, ybase, true, ypre2, false, mk, dummy, dcost);
    } else {
      z_row<T, S, B, C, false>(A, xs, zs, wv, R0, CJ0, lane, ybase, true, ypre, true, mk, zown, cost_data);
      if (has_z_halo) z_row<T, S, B, C, false>(A, xs, zs, hrowz, R0, CJ0, lane, ybase, true, ypre2, false, mk, dummy, dcost);
    }
  }
  };
  if (!SRMAP_EXP_REGFIRST) phase1_data();
  // ---------------- phase 1: regulariser ----------------
  if (want_reg) {
    const bool reg_border = (R0 + C::TH + C::WIN > A.H) || (C0 + C::TW + C::WIN > A.W);
    const bool cost_row = gr >= A.cr0 && gr < A.cr1;
    if (reg_border)
      reg_row<T, S, REGK, R, C, true, true>(acc, cost_reg, xs, cs, wreg, wv, lane, gr, gc0, A.W, A.H, A.lambda, A.powtab, A.pwsum, cost_row);
    else
      reg_row<T, S, REGK, R, C, false, true>(acc, cost_reg, xs, cs, wreg, wv, lane, gr, gc0, A.W, A.H, A.lambda, A.powtab, A.pwsum, cost_row);
    if (has_reg_halo) {
      T dacc[S];
      double dc = 0.0;
      T whl[S];
#pragma unroll
      for (int pc = 0; pc < S; ++pc) whl[pc] = whs[((wv - 2) * S + pc) * C::CW + lane];
      // right-edge masks follow the tile's; the rows above a tile reach below the image only when the tile keeps
      // fewer than WIN rows of it (a partial bottom tile: found by tests/test_gpu_fuzz.py, H - R0 = 2 with BTV(3))
      if (C0 + C::TW + C::WIN > A.W || R0 + C::WIN > A.H)
        reg_row<T, S, REGK, R, C, true, false>(dacc, dc, xs, cs, whl, hrow, lane, R0 + hrow, gc0, A.W, A.H, A.lambda, A.powtab, A.pwsum, false);
      else
        reg_row<T, S, REGK, R, C, false, false>(dacc, dc, xs, cs, whl, hrow, lane, R0 + hrow, gc0, A.W, A.H, A.lambda, A.powtab, A.pwsum, false);
    }
    // left halo columns -1 .. -RU, one row per lane, one column per wave (a few-lane task with a
    // long dependent chain -- both columns on one wave made the whole workgroup wait for it at the barrier)
    if (col_task) {  // waves 4 / 5: their SIMDs carry the z halo rows only
      const T wcol = wcs[(wv - 4) * 16 + lane];
      const int rowrel = lane - RU;
      const int lo = rowrel * C::XROW, lc = rowrel * C::CROW;  // per-lane row offsets
      if (reg_border) {
        if (RU >= 1 && wv == 4) reg_halo_col<T, S, REGK, R, C, -1, true>(xs + lo, cs + lc, wcol, rowrel, R0, C0, A.W, A.H, A.lambda, A.powtab);
        if (RU >= 2 && wv == 5) reg_halo_col<T, S, REGK, R, C, -2, true>(xs + lo, cs + lc, wcol, rowrel, R0, C0, A.W, A.H, A.lambda, A.powtab);
      } else {
        if (RU >= 1 && wv == 4) reg_halo_col<T, S, REGK, R, C, -1, false>(xs + lo, cs + lc, wcol, rowrel, R0, C0, A.W, A.H, A.lambda, A.powtab);
        if (RU >= 2 && wv == 5) reg_halo_col<T, S, REGK, R, C, -2, false>(xs + lo, cs + lc, wcol, rowrel, R0, C0, A.W, A.H, A.lambda, A.powtab);
      }
    }
  }
  if (SRMAP_EXP_REGFIRST) phase1_data();
  __syncthreads();

  // ---------------- phase 2 ----------------
  T dreg[S];  // WD: the search direction at this thread's pixels (g.d is produced with g)
#pragma unroll
  for (int pc = 0; pc < S; ++pc) dreg[pc] = T(0);
  if (WD && gr < A.H && gc0 < A.W && gr >= A.cr0 && gr < A.cr1) {
    const T* dp = A.dvec + (size_t)ch * N + (size_t)gr * A.W + gc0;
#pragma unroll
    for (int pc = 0; pc < S; ++pc) dreg[pc] = dir_elem<T>(dp[pc], dsc);
  }
  if (want_data && A.g != nullptr) {
    const T sc = (T)(2 * S * S);  // g += 2 * (s*s block sum) (objective_data_term.cpp:55-71)
#pragma unroll
    for (int pc = 0; pc < S; ++pc) {
      T zz;
      if (B == 1) {
        zz = zown[pc];
      } else {
        zz = T(0);
#pragma unroll
        for (int a = 0; a < B; ++a) zz += k1_tap<B>(A, a) * zs[(wv + a) * C::ZROW + pc * C::CW + lane];  // rows wv-HB+a
      }
      if (SP)  // the ring pass owns the pixels within Dr of the image edge
        zz = ((unsigned)(gr - A.Dr) < (unsigned)(A.H - 2 * A.Dr) && (unsigned)(gc0 + pc - A.Dr) < (unsigned)(A.W - 2 * A.Dr)) ? zz : T(0);
      acc[pc] += sc * zz;
    }
  }
  if (want_reg && A.g != nullptr) reg_pass2z<T, S, REGK, R, C>(acc, xs, cs, wv, lane, A.powtab);

  if (SP && want_data && A.g != nullptr && A.ringbuf != nullptr) {
    // the ring pass ran AHEAD of this launch (k_gather_ring into the plan's buffer): its exact data gradient of the pixels
    // within Dr of the image edge is added LAST, as the pass added it to the stored g when it ran behind the launch
    // (uniform test first: only edge tiles hold such pixels)
    const int Dr = A.Dr;
    if (R0 < Dr || R0 + C::TH > A.H - Dr || C0 < Dr || C0 + C::TW > A.W - Dr) {
      const int band = Dr * A.W, mid = A.H - 2 * Dr;
      const T* rb = A.ringbuf + (size_t)ch * (size_t)(2 * band + 2 * Dr * mid);
#pragma unroll
      for (int pc = 0; pc < S; ++pc) {
        const int gc = gc0 + pc;
        if (gr < A.H && gc < A.W) {
          int t = -1;
          if (gr < Dr) t = gr * A.W + gc;
          else if (gr >= A.H - Dr) t = band + (gr - (A.H - Dr)) * A.W + gc;
          else if (gc < Dr) t = 2 * band + (gr - Dr) * 2 * Dr + gc;
          else if (gc >= A.W - Dr) t = 2 * band + (gr - Dr) * 2 * Dr + Dr + (gc - (A.W - Dr));
          if (t >= 0) acc[pc] += rb[t];
        }
      }
    }
  }
  if (A.g != nullptr && gr < A.H && gc0 < A.W) {
    T* dst = A.g + (size_t)ch * N + (size_t)gr * A.W + gc0;
#pragma unroll
    for (int pc = 0; pc < S; ++pc) __builtin_nontemporal_store(acc[pc], &dst[pc]);  // written once, not re-read here
  }

  // ---------------- cost partial of this workgroup ----------------
  {
    double gd = 0.0;
    if (WD) {
#pragma unroll
      for (int pc = 0; pc < S; ++pc) gd += (double)acc[pc] * (double)dreg[pc];
      gd = wave_sum_d(gd);
    }
    const double cw = wave_sum_d((double)(S * S) * cost_data + cost_reg);
    if (lane == 0) { red[0][wv] = cw; if (WD) red[1][wv] = gd; }
    __syncthreads();
    if (tid == 0) {
      double c = 0.0, d = 0.0;
#pragma unroll
      for (int i = 0; i < C::NW; ++i) { c += red[0][i]; if (WD) d += red[1][i]; }
      const size_t b = ((size_t)blockIdx.z * nby_t + by) * gridDim.x + blockIdx.x;
      put_partial<WD>(A, b, c, d);
    }
    // in-kernel finish: the last workgroup of the grid gathers the granules of the evaluation
    if (A.mfinish && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && blockIdx.z == gridDim.z - 1) {
      __syncthreads();
      finish_block<WD, C::NT>(A, &red[0][0]);
    }
  }
}

// After the tile kernel: subtract the border corrections from g and reduce every cost partial of the evaluation
// in index order (deterministic).  Block 0 reduces; all blocks apply corrections.
template <typename T>
__global__ __launch_bounds__(256) void k_finish_eval(T* __restrict__ g, const T* __restrict__ corr, int n_ring, int W,
                                                     int H, RingRects ring, int C, const double* __restrict__ partials,
                                                     int n_partials, double* __restrict__ cost_out,
                                                     const double* __restrict__ partials_gd, double* pub,
                                                     double* tag_slot, double tag) {
  __shared__ double red[4], red2[4];
  double v = 0.0, v2 = 0.0;
  if (blockIdx.x == 0) {
    // eight requests in flight per thread (one load per iteration was one memory round trip per 256 partials);
    // the order of the additions is fixed
    constexpr int U = 8;
    for (int base = threadIdx.x; base < n_partials; base += 256 * U) {
      double a[U], b2[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * 256;
        a[u] = i < n_partials ? partials[i] : 0.0;
        b2[u] = (partials_gd != nullptr && i < n_partials) ? partials_gd[i] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { v += a[u]; v2 += b2[u]; }
    }
  }
  // block 0 only reduces (its chain of dependent steps is the kernel's duration); the corrections start at block 1
  const int t = ((int)blockIdx.x - 1) * 256 + (int)threadIdx.x;
  if (g != nullptr && blockIdx.x > 0 && t < n_ring) {
    int qr, qc;
    ring_pixel(t, W, H, ring, qr, qc);
    if (qr >= 0 && qr < H && qc >= 0 && qc < W) {
      for (int ch = 0; ch < C; ++ch) {
        const T c = corr[(size_t)ch * n_ring + t];
        if (c != T(0)) g[((size_t)ch * H + qr) * W + qc] -= c;
      }
    }
  }
  if (blockIdx.x == 0) {
    v = wave_sum_d(v);
    if (partials_gd != nullptr) v2 = wave_sum_d(v2);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { red[wid] = v; red2[wid] = v2; }
    __syncthreads();
    if (threadIdx.x == 0) cost_out[0] = (red[0] + red[1]) + (red[2] + red[3]);
    if (partials_gd != nullptr) {  // g.d of the same evaluation (solver line search)
      if (threadIdx.x == 0) {
        const double gd = (red2[0] + red2[1]) + (red2[2] + red2[3]);
        cost_out[1] = gd;
        if (pub != nullptr) {  // solver line search: {cost, g.d} straight to the host-mapped words, then the arrival tag
          pub[0] = cost_out[0];
          pub[1] = gd;
          __threadfence_system();
          *(volatile double*)tag_slot = tag;
        }
      }
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// host side: plan (per problem, owned by the problem) and launch

// Up to this many cost partials (tiles + border blocks of all channels) one workgroup reduces them -- the tile kernel's
// last one, or k_finish_eval's block 0 -- and the solver gets g.d out of the same launch; beyond it the caller reduces in
// two stages.  64 K covers configs[2] (RGB 4096^2: 24 576 tiles), which at 16 K paid two more launches per evaluation and
// a separate 400 MB dot-product pass per trial point of a solve.
constexpr size_t kMaxFusedPartials = 65536;

void ztile_release(srmap_problem* p) {
  ZPlan* z = static_cast<ZPlan*>(p->zplan);
  if (!z) return;
  if (z->d_hdr) (void)hipFree(z->d_hdr);
  if (z->d_ent) (void)hipFree(z->d_ent);
  if (z->d_cnt) (void)hipFree(z->d_cnt);
  if (z->d_off) (void)hipFree(z->d_off);
  if (z->d_aux) (void)hipFree(z->d_aux);
  if (z->d_spw) (void)hipFree(z->d_spw);
  if (z->d_spsrc) (void)hipFree(z->d_spsrc);
  if (z->d_ringbuf) (void)hipFree(z->d_ringbuf);
  spfwd_release(&z->spf);
  if (z->d_corr) (void)hipFree(z->d_corr);
  if (z->d_bd) (void)hipFree(z->d_bd);
  if (z->d_mpart) (void)hipFree(z->d_mpart);
  delete z;
  p->zplan = nullptr;
}

void ztile_rearm(srmap_problem* p) {
  ZPlan* z = static_cast<ZPlan*>(p->zplan);
  if (z && z->d_mpart) (void)hipMemsetD32((hipDeviceptr_t)z->d_mpart, (int)kSentinel32, 4 * z->mpart_cap);
}

// Decide whether k_eval_z covers the problem; build the frame table.
bool ztile_plan(srmap_problem* p) {
  ztile_release(p);
  const Geometry& g = p->geo;
  const int S = g.s, B = g.b, K = g.K;
  if (!p->maps_regular) return false;
  if (S < 2 || S > 4) return false;
  if (B != 1 && B != 3) return false;
  if (B == 3) {  // the tile kernel carries the three distinct values of the symmetric Gaussian kernel
    const std::vector<double>& k2 = p->blur2d;
    if (!(k2[0] == k2[2] && k2[0] == k2[6] && k2[0] == k2[8] && k2[1] == k2[3] && k2[1] == k2[5] && k2[1] == k2[7] &&
          p->blur1d[0] == p->blur1d[2]))
      return false;
  }
  std::vector<int> ox(K, 0), oy(K, 0);
  int amax = 0;
  bool subpix = false;
  if (p->has_motion) {
    for (int k = 0; k < K; ++k) {
      const WarpTaps<double>& f = p->fwd_warps[k];
      const WarpTaps<double>& b = p->bwd_warps[k];
      if (f.ytab != nullptr || b.ytab != nullptr) return false;  // rounding-tie shifts: per-row tables, direct kernels
      if (f.ntaps != 1 || b.ntaps != 1) subpix = true;
      else if (b.ox != -f.ox || b.oy != -f.oy) return false;
      ox[k] = f.ox; oy[k] = f.oy;
      amax = std::max(amax, std::max(std::abs(f.ox), std::abs(f.oy)));
      amax = std::max(amax, std::max(std::abs(b.ox), std::abs(b.oy)));
    }
  }
  if (amax > 4096) return false;
  if ((long long)g.W * g.H >= (long long)INT_MAX) return false;  // the border blocks index a plane with 32 bits
  ZPlan* z = new ZPlan();
  z->S = S; z->B = B;
  z->E = amax;
  if (subpix) {
    // ---- sub-pixel plan: z(p) = sum over (frame, bilinear tap of the TRANSPOSE warp) pairs that land on the LR grid ----
    z->subpix = true;
    z->E = 0;  // no border blocks: the forward kernel counts the cost, the ring pass evaluates the border exactly
    z->Dr = amax + 2 + g.hb;
    if (g.W <= 4 * z->Dr + 2 * S || g.H <= 4 * z->Dr + 2 * S) { delete z; return false; }
    for (int r = 0; r < p->nreg && z->regk == 0; ++r) {
      const RegSpec& rs = p->reg[r];
      if (rs.lambda <= 0) continue;
      if (rs.kind == SRMAP_REG_TV) { z->regk = 1; z->reg_index = r; }
      else if (rs.kind == SRMAP_REG_BTV && rs.range >= 1 && rs.range <= 3) { z->regk = 2; z->regr = rs.range; z->reg_index = r; }
      break;
    }
    struct SpE { ZEntry e; double w; };
    std::vector<std::vector<SpE>> lists((size_t)S * S);
    for (int pr = 0; pr < S; ++pr)
      for (int pc = 0; pc < S; ++pc)
        for (int k = 0; k < K; ++k) {
          const WarpTaps<double>& b = p->bwd_warps[k];
          for (int t = 0; t < b.ntaps; ++t) {  // t & 1 = dx tap, t >> 1 = dy tap (motion_module.cpp:40-51, gather form)
            if (b.w[t] == 0.0) continue;
            const int rr = pr + b.oy + (t >> 1), cc = pc + b.ox + (t & 1);
            if (pmod(rr, S) != 0 || pmod(cc, S) != 0) continue;
            SpE q; q.e.k = k; q.e.io = fdiv(rr, S); q.e.jo = fdiv(cc, S); q.e.oyx = 0; q.w = b.w[t];
            lists[(size_t)pr * S + pc].push_back(q);
          }
        }
    // the same taps source-major (z_row_sp2): per row phase one record per (frame, vertical tap) on the LR grid
    std::vector<std::vector<ZSrc>> srcs((size_t)S);
    for (int pr = 0; pr < S; ++pr)
      for (int k = 0; k < K; ++k) {
        const WarpTaps<double>& b = p->bwd_warps[k];
        for (int dy = 0; dy < 2; ++dy) {
          double w0 = 0.0, w1 = 0.0;
          for (int t = 0; t < b.ntaps; ++t)
            if ((t >> 1) == dy) { if (t & 1) w1 = b.w[t]; else w0 = b.w[t]; }
          if (w0 == 0.0 && w1 == 0.0) continue;
          const int rr = pr + b.oy + dy;
          if (pmod(rr, S) != 0) continue;
          ZSrc q; q.k = k; q.io = fdiv(rr, S); q.w0 = w0; q.w1 = w1;
          const int a = pmod(-b.ox, S);
          q.q = (b.ox + a) / S;  // exact
          // the first residual column e in {-1, 0, 1} that a cell's S + 2 HB pixels use (sp_apply derives the same from a)
          const bool lo = (a - S >= -g.hb) || (a - 1 - S >= -g.hb);
          q.am = a | ((lo ? 0 : 1) << 8);  // (e_lo + 1) << 8
          srcs[(size_t)pr].push_back(q);
        }
      }
    int spmax = kSpChunk;
    for (const auto& l : srcs) spmax = std::max(spmax, (int)((l.size() + kSpChunk - 1) / kSpChunk) * kSpChunk);
    z->spmax = spmax;
    std::vector<ZSrc> srctab((size_t)S * spmax, ZSrc{0, 0, 0, kSpNull, 0.0, 0.0});   // null records behind the sources
    for (int pr = 0; pr < S; ++pr) {
      z->spn[pr] = (int)((srcs[(size_t)pr].size() + kSpChunk - 1) / kSpChunk) * kSpChunk;
      for (size_t n = 0; n < srcs[(size_t)pr].size(); ++n) srctab[(size_t)pr * spmax + n] = srcs[(size_t)pr][n];
    }
    int MS = 1;
    for (const auto& l : lists) MS = std::max(MS, (int)l.size());
    z->MS = MS;
    std::vector<int> cnt((size_t)S * 8, 0);
    std::vector<ZEntry> aux((size_t)S * MS * S, ZEntry{0, 0, 0, 0});
    std::vector<double> spw((size_t)S * MS * S, 0.0);
    for (int pr = 0; pr < S; ++pr) {
      int mx = 0;
      for (int pc = 0; pc < S; ++pc) {
        const auto& l = lists[(size_t)pr * S + pc];
        cnt[(size_t)pr * 8 + pc] = (int)l.size();
        mx = std::max(mx, (int)l.size());
        for (size_t t = 0; t < l.size(); ++t) {
          const size_t slot = ((size_t)t * S + pr) * S + pc;
          aux[slot] = l[t].e;
          spw[slot] = l[t].w;
        }
      }
      cnt[(size_t)pr * 8 + S] = mx;
    }
    for (size_t i = 0; i < cnt.size(); ++i) z->h_cnt[i] = cnt[i];
    bool ok = hipMalloc((void**)&z->d_cnt, sizeof(int) * cnt.size()) == hipSuccess &&
              hipMemcpy(z->d_cnt, cnt.data(), sizeof(int) * cnt.size(), hipMemcpyHostToDevice) == hipSuccess &&
              hipMalloc((void**)&z->d_aux, sizeof(ZEntry) * aux.size()) == hipSuccess &&
              hipMemcpy(z->d_aux, aux.data(), sizeof(ZEntry) * aux.size(), hipMemcpyHostToDevice) == hipSuccess &&
              hipMalloc((void**)&z->d_spw, sizeof(double) * spw.size()) == hipSuccess &&
              hipMemcpy(z->d_spw, spw.data(), sizeof(double) * spw.size(), hipMemcpyHostToDevice) == hipSuccess &&
              hipMalloc((void**)&z->d_spsrc, sizeof(ZSrc) * srctab.size()) == hipSuccess &&
              hipMemcpy(z->d_spsrc, srctab.data(), sizeof(ZSrc) * srctab.size(), hipMemcpyHostToDevice) == hipSuccess &&
              hipMalloc((void**)&z->d_off, 64) == hipSuccess;
    p->zplan = z;
    if (ok && gather_ring_kernel_ok(p, g, K, z->Dr)) {
      // the ring pass runs ahead of the tile kernel into this buffer (g.d and the cost then finish inside the tile launch)
      const size_t nring = 2 * (size_t)z->Dr * g.W + 2 * (size_t)z->Dr * (g.H - 2 * z->Dr);
      ok = hipMalloc(&z->d_ringbuf, nring * g.C * p->elem()) == hipSuccess;
    }
    if (ok) {  // granules of the in-kernel cost reduction (as below)
      const size_t cap = std::min<size_t>(ztile_partials_needed(p), kMaxFusedPartials);
      ok = hipMalloc((void**)&z->d_mpart, 2 * cap * sizeof(double)) == hipSuccess &&
           hipMemsetD32((hipDeviceptr_t)z->d_mpart, (int)kSentinel32, 4 * cap) == hipSuccess;
      z->mpart_cap = cap;
    }
    if (!ok) { ztile_release(p); return false; }
    (void)spfwd_plan(p, &z->spf);
    return true;
  }
  // the border frame (k_border, width 2E) must stay a small part of the image
  if (g.W <= 4 * z->E + 2 * S || g.H <= 4 * z->E + 2 * S) { delete z; return false; }
  // the one regulariser handled in-kernel (first TV / BTV with lambda > 0)
  for (int r = 0; r < p->nreg && z->regk == 0; ++r) {
    const RegSpec& rs = p->reg[r];
    if (rs.lambda <= 0) continue;
    if (rs.kind == SRMAP_REG_TV) { z->regk = 1; z->reg_index = r; }
    else if (rs.kind == SRMAP_REG_BTV && rs.range >= 1 && rs.range <= 3) { z->regk = 2; z->regr = rs.range; z->reg_index = r; }
    break;  // only the first active regulariser may be fused (order of accumulation)
  }
  // frame table: pixel (row phase pr, column phase pc) owns the residual of frame k iff (pr - oy) and (pc - ox)
  // are multiples of S; the LR pixel is (rc + io, cell + jo)
  std::vector<int2> hdr((size_t)S * S);
  std::vector<ZEntry> ent;
  for (int pr = 0; pr < S; ++pr)
    for (int pc = 0; pc < S; ++pc) {
      int2 h; h.x = 0; h.y = (int)ent.size();
      for (int k = 0; k < K; ++k) {
        if (pmod(pr - oy[k], S) != 0 || pmod(pc - ox[k], S) != 0) continue;
        ZEntry e; e.k = k; e.io = fdiv(pr - oy[k], S); e.jo = fdiv(pc - ox[k], S);
        e.oyx = (int)(((unsigned)oy[k] << 16) | ((unsigned)ox[k] & 0xffffu));
        ent.push_back(e);
        h.x++;
      }
      hdr[(size_t)pr * S + pc] = h;
    }
  if (ent.empty()) { ZEntry e = {0, 0, 0, 0}; ent.push_back(e); }
  if ((int)ent.size() > kBorderTabEntries) { delete z; return false; }  // k_border stages the table in LDS
  z->n_ent = (int)ent.size();
  // the tile kernel's layout of the same table: [pr][t][pc], with the observation's element offset precomputed
  int MS = 1;
  for (const int2& h : hdr) MS = std::max(MS, h.x);
  z->MS = MS;
  std::vector<int> cnt((size_t)S * 8, 0);
  std::vector<long long> off((size_t)S * MS * S, 0);
  std::vector<ZEntry> aux((size_t)S * MS * S, ZEntry{0, 0, 0, 0});
  const long long nl = (long long)g.w * g.h;
  for (int pr = 0; pr < S; ++pr) {
    int mx = 0;
    for (int pc = 0; pc < S; ++pc) {
      const int2 h = hdr[(size_t)pr * S + pc];
      cnt[(size_t)pr * 8 + pc] = h.x;
      mx = std::max(mx, h.x);
      for (int t = 0; t < h.x; ++t) {
        const ZEntry& e = ent[(size_t)h.y + t];
        const size_t slot = ((size_t)t * S + pr) * S + pc;
        aux[slot] = e;
        off[slot] = (long long)e.k * g.C * nl + (long long)e.io * g.w + e.jo;
      }
    }
    cnt[(size_t)pr * 8 + S] = mx;
    int mn = INT_MAX;
    for (int pc = 0; pc < S; ++pc) mn = std::min(mn, cnt[(size_t)pr * 8 + pc]);
    cnt[(size_t)pr * 8 + S + 1] = mn;
    for (int pc = 0; pc < S; ++pc) {
      z->h_off0[pr * 4 + pc] = off[((size_t)0 * S + pr) * S + pc];
      z->h_aux0[pr * 4 + pc] = aux[((size_t)0 * S + pr) * S + pc];
    }
  }
  for (size_t i = 0; i < cnt.size(); ++i) z->h_cnt[i] = cnt[i];
  bool ok = hipMalloc((void**)&z->d_hdr, sizeof(int2) * hdr.size()) == hipSuccess &&
            hipMemcpy(z->d_hdr, hdr.data(), sizeof(int2) * hdr.size(), hipMemcpyHostToDevice) == hipSuccess &&
            hipMalloc((void**)&z->d_ent, sizeof(ZEntry) * ent.size()) == hipSuccess &&
            hipMemcpy(z->d_ent, ent.data(), sizeof(ZEntry) * ent.size(), hipMemcpyHostToDevice) == hipSuccess &&
            hipMalloc((void**)&z->d_cnt, sizeof(int) * cnt.size()) == hipSuccess &&
            hipMemcpy(z->d_cnt, cnt.data(), sizeof(int) * cnt.size(), hipMemcpyHostToDevice) == hipSuccess &&
            hipMalloc((void**)&z->d_off, sizeof(long long) * off.size()) == hipSuccess &&
            hipMemcpy(z->d_off, off.data(), sizeof(long long) * off.size(), hipMemcpyHostToDevice) == hipSuccess &&
            hipMalloc((void**)&z->d_aux, sizeof(ZEntry) * aux.size()) == hipSuccess &&
            hipMemcpy(z->d_aux, aux.data(), sizeof(ZEntry) * aux.size(), hipMemcpyHostToDevice) == hipSuccess;
  if (ok && z->E > 0) {
    // the rectangles of the border frame that can carry work (ztile_dev.hpp RingRects)
    int mnx = 0, mxx = 0, mny = 0, mxy = 0;
    for (int k = 0; k < K; ++k) {
      mnx = std::min(mnx, ox[k]); mxx = std::max(mxx, ox[k]);
      mny = std::min(mny, oy[k]); mxy = std::max(mxy, oy[k]);
    }
    RingRects rr;
    rr.rg[0] = (B > 1) ? std::max(0, mxy) : 0;   // corrections need a blur tap between the pixel and the residual
    rr.rg[1] = (B > 1) ? std::max(0, mxx) : 0;
    rr.rg[2] = std::max(0, -mny);
    rr.rg[3] = std::max(0, mxy - S + 1);
    rr.rg[4] = std::max(0, -mnx);
    rr.rg[5] = std::max(0, mxx - S + 1);
    z->ring = rr;
    z->n_ring = (int)ring_count(rr, g.W, g.H);
    if (z->n_ring > 0) ok = hipMalloc(&z->d_corr, (size_t)g.C * z->n_ring * p->elem()) == hipSuccess;
  }
  if (ok) {
    auto put = [&](auto bd) {
      bd.hdr = z->d_hdr; bd.ent = z->d_ent; bd.blur_d = (decltype(bd.blur_d))p->d_blur; bd.corr = (decltype(bd.corr))z->d_corr;
      bd.S = S; bd.b = B; bd.hb = g.hb; bd.n_ring = z->n_ring; bd.n_ent = z->n_ent; bd.obs_C = g.C;
      return hipMalloc(&z->d_bd, sizeof(bd)) == hipSuccess &&
             hipMemcpy(z->d_bd, &bd, sizeof(bd), hipMemcpyHostToDevice) == hipSuccess;
    };
    ok = p->dtype == SRMAP_F32 ? put(BorderArgs<float>()) : put(BorderArgs<double>());
  }
  p->zplan = z;
  if (ok) {
    // granules of the in-kernel cost reduction (one per tile + one per border block), up to a cap beyond which the
    // caller's two-stage reduction is used anyway
    const size_t cap = std::min<size_t>(ztile_partials_needed(p), kMaxFusedPartials);
    ok = hipMalloc((void**)&z->d_mpart, 2 * cap * sizeof(double)) == hipSuccess &&
         hipMemsetD32((hipDeviceptr_t)z->d_mpart, (int)kSentinel32, 4 * cap) == hipSuccess;
    z->mpart_cap = cap;
  }
  if (!ok) { ztile_release(p); return false; }
  return true;  // the caller preloads the kernel instance (ztile_preload)
}

// Frame sharding may split the regulariser by row band when the tile kernel alone produces it: one active
// regulariser, the one fused into the plan (no 3-D TV / second regulariser / wide BTV pass of the direct kernels).
bool ztile_overlaps_halo(const srmap_problem* p) {
  const ZPlan* z = static_cast<const ZPlan*>(p->zplan);
  return z != nullptr && p->impl != SRMAP_IMPL_DIRECT && !z->subpix;
}

bool ztile_reg_band_ok(const srmap_problem* p, unsigned terms) {
  const ZPlan* z = static_cast<const ZPlan*>(p->zplan);
  if (!z || p->impl == SRMAP_IMPL_DIRECT) return false;
  if (!(terms & SRMAP_TERM_REG)) return false;
  int active = 0;
  for (int r = 0; r < p->nreg; ++r)
    if (p->reg[r].lambda > 0.0) { if (!(z->regk != 0 && r == z->reg_index)) return false; active++; }
  return active == 1;
}

// Upper bound of the cost partials one tile launch over C channels produces (tiles + border blocks, which fill whole
// grid rows) -- ONE definition for the granule allocation, the launch and the solver's decisions.
static size_t ztile_est_partials(const ZPlan* z, int w, int H, int C) {
  const size_t tiles = (size_t)((w + 63) / 64) * ((H + 7) / 8) * C;
  const size_t ring = (z && z->n_ring > 0) ? (size_t)((z->n_ring + 511) / 512 + (H + 7) / 8) * C : (size_t)0;
  return tiles + ring;
}

// Whether the tile launch of an evaluation with `terms` over C channels is the g.d instance (k_eval_z<..., WD>): it must
// produce the WHOLE gradient (no further regulariser kernel behind it) and few enough partials for its in-kernel
// reduction.  launch_eval_ztile asks it for the launch, ztile_can_fold for the solver (a folded trial point exists only
// inside that instance): the two cannot drift apart.
static bool ztile_gd_instance_ok(const srmap_problem* p, const ZPlan& z, int w, int H, int C, unsigned terms) {
  // sub-pixel plans: the ring pass must be able to run AHEAD of the tile launch (into z.d_ringbuf) -- a pass that adds to g
  // behind it would leave the launch's g.d without the ring's share
  if (z.subpix && z.d_ringbuf == nullptr) return false;
  if (terms & SRMAP_TERM_REG)
    for (int r = 0; r < p->nreg; ++r)
      if (!(z.regk != 0 && r == z.reg_index) && p->reg[r].lambda > 0.0) return false;
  return ztile_est_partials(&z, w, H, C) <= kMaxFusedPartials;
}

bool ztile_can_fold(const srmap_problem* p) {
  const ZPlan* z = static_cast<const ZPlan*>(p->zplan);
  if (!z || p->impl == SRMAP_IMPL_DIRECT || p->ov_hook != nullptr) return false;
  const Geometry& g = p->geo;
  // sub-pixel plans: the trial point is formed by the forward tile kernel, whose window has to hold a workgroup's own pixels
  if (z->subpix && !(z->spf.ok && z->spf.can_fold)) return false;
  return ztile_gd_instance_ok(p, *z, g.w, g.H, p->view_C > 0 ? p->view_C : g.C, SRMAP_TERM_ALL);
}

size_t ztile_partials_needed(const srmap_problem* p) {
  const Geometry& g = p->geo;
  return ztile_est_partials(static_cast<const ZPlan*>(p->zplan), g.w, g.H, g.C);
}

template <typename T, int S, int B, int REGK, int R>
static int launch_z(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g,
                    const T* wts, const ZPlan& z, double* partials, int* nblocks, hipStream_t st, const T* dvec,
                    double* partials_gd, MFin mfin, bool ring_ahead) {
  using C = ZCfg<T, S, B, REGK, R>;
  ZArgs<T, B, C::NP> A;
  fill_zargs<T, S, B, REGK, R>(A, p, geo, obs_c0, terms, x, g, wts, z, partials, dvec, partials_gd);
  if (dvec != nullptr && p->eval_fold_xk != nullptr) {  // line search: the point is xk + stp * d; its own pixels go to x
    A.fold_norms = p->eval_fold_norms;
    if (!(z.subpix && (terms & SRMAP_TERM_DATA))) {
      A.fold_xk = (const T*)p->eval_fold_xk; A.fold_x = const_cast<T*>(x); A.fold_stp = (T)p->eval_fold_stp;
    }  // sub-pixel plans: the forward kernel ahead of this launch formed the point and wrote x (launch_eval_ztile); dvec is
       // still the unnormalised direction here (fold_norms)
  }
  dim3 grid((geo.w + C::CW - 1) / C::CW, (geo.H + C::TH - 1) / C::TH, geo.C);
  { const unsigned t = grid.x; grid.x = grid.y; grid.y = t; }
  const int n_tile_partials = (int)(grid.x * grid.y * grid.z);
  // border blocks: whole rows of the grid in front of the tiles
  A.bd = (const BorderArgs<T>*)z.d_bd;
  A.nby = 0; A.n_tile_partials = n_tile_partials;
  int nbb = 0;
  if ((terms & SRMAP_TERM_DATA) && z.n_ring > 0) {
    const int need = (z.n_ring + C::NT - 1) / C::NT;
    A.nby = (need + (int)grid.x - 1) / (int)grid.x;
    nbb = A.nby * (int)grid.x;
    grid.y += A.nby;
  }
  A.rbuf = nullptr; A.spw = z.d_spw; A.Dr = z.Dr; A.ringbuf = nullptr;

  A.mfinish = mfin.on ? 1 : 0;
  A.n_partials = n_tile_partials + nbb * (int)grid.z;
  A.mpart = z.d_mpart;
  A.mpart_gd = z.d_mpart ? z.d_mpart + z.mpart_cap : nullptr;
  A.cost_out = p->d_cost;
  A.pub = mfin.publish ? p->eval_pub : nullptr;
  A.tag_slot = p->eval_pub_tag_slot;
  A.tag = p->eval_pub_tag;
  A.to_host = p->eval_timeout_host;
  A.xpart = mfin.xpart; A.n_xpart = mfin.n_xpart;
  if (mfin.on && (size_t)A.n_partials > z.mpart_cap) return set_error(p->ctx, SRMAP_EHIP, "granule capacity");
  A.sel_mode = 0; A.sel0 = 0; A.sel1 = 0;
  if (z.subpix && (terms & SRMAP_TERM_DATA)) {
    A.rbuf = (const T*)p->d_resid;
    A.obs_C = geo.C;  // layout of the residual buffer written by launch_forward_direct for this evaluation
    A.ringbuf = ring_ahead ? (const T*)z.d_ringbuf : nullptr;
    if (dvec != nullptr) hipLaunchKernelGGL((k_eval_z<T, S, B, REGK, R, true, true>), grid, dim3(C::NT), SRMAP_EXP_LDS_PAD, st, A);
    else hipLaunchKernelGGL((k_eval_z<T, S, B, REGK, R, false, true>), grid, dim3(C::NT), SRMAP_EXP_LDS_PAD, st, A);
  } else {
    auto launch = [&]() {
      if (dvec != nullptr) hipLaunchKernelGGL((k_eval_z<T, S, B, REGK, R, true, false>), grid, dim3(C::NT), SRMAP_EXP_LDS_PAD, st, A);
      else hipLaunchKernelGGL((k_eval_z<T, S, B, REGK, R, false, false>), grid, dim3(C::NT), SRMAP_EXP_LDS_PAD, st, A);
    };
    if (p->ov_hook != nullptr) {
      // Row shard: a tile row [8 t, 8 t + 8) reads x rows within the halo width of itself, so the tile rows t with
      // 8 t >= 2 * ov_top and 8 t + 8 <= H - 2 * ov_bot touch no halo row.  They run first, the halo exchange is
      // posted on its own stream under them, and the boundary tile rows (and the border blocks) follow the event.
      const int th = C::TH;
      A.sel0 = (2 * p->ov_top + th - 1) / th;
      A.sel1 = (geo.H - 2 * p->ov_bot) / th;
      if (A.sel1 > A.sel0) { A.sel_mode = 1; launch(); }
      const int rch = p->ov_hook(p->ov_arg);
      if (rch) return rch;
      SRMAP_HIP(p->ctx, hipStreamWaitEvent(st, p->ov_event, 0));
      if (A.sel1 > A.sel0) { A.sel_mode = 2; launch(); } else { A.sel_mode = 0; launch(); }
    } else {
      launch();
    }
  }
  *nblocks = n_tile_partials + nbb * (int)grid.z;
  SRMAP_HIP(p->ctx, hipGetLastError());
  return SRMAP_OK;
}

// HIP loads a kernel's code object lazily at its first launch (milliseconds): touch the instance when the plan is
// made, not inside the first evaluation of a solve.
template <typename T, int S, int B, int REGK, int R>
static void preload_z() {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_eval_z<T, S, B, REGK, R, false, false>));
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_eval_z<T, S, B, REGK, R, true, false>));
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_eval_z<T, S, B, REGK, R, false, true>));
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_eval_z<T, S, B, REGK, R, true, true>));
}
template <typename T, int S, int B>
static void preload_reg(int regk, int regr) {
  preload_z<T, S, B, 0, 0>();
  if (regk == 1) preload_z<T, S, B, 1, 0>();
  if (regk == 2 && regr == 1) preload_z<T, S, B, 2, 1>();
  if (regk == 2 && regr == 2) preload_z<T, S, B, 2, 2>();
  if (regk == 2 && regr == 3) preload_z<T, S, B, 2, 3>();
}
template <typename T>
static void preload_sb(int S, int B, int regk, int regr) {
#ifdef SRMAP_ZT_ONLY_CFG2
  if (sizeof(T) == sizeof(SRMAP_ZT_ONLY_T) && S == 4 && B == SRMAP_ZT_ONLY_B && regk == 2 && regr == 3) preload_z<SRMAP_ZT_ONLY_T, 4, SRMAP_ZT_ONLY_B, 2, 3>();
  return;
#else
  if (S == 2 && B == 1) preload_reg<T, 2, 1>(regk, regr);
  else if (S == 2 && B == 3) preload_reg<T, 2, 3>(regk, regr);
  else if (S == 3 && B == 1) preload_reg<T, 3, 1>(regk, regr);
  else if (S == 3 && B == 3) preload_reg<T, 3, 3>(regk, regr);
  else if (S == 4 && B == 1) preload_reg<T, 4, 1>(regk, regr);
  else if (S == 4 && B == 3) preload_reg<T, 4, 3>(regk, regr);
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_finish_eval<T>));
#endif
}
void ztile_preload(const srmap_problem* p) {
  const ZPlan* z = static_cast<const ZPlan*>(p->zplan);
  if (!z) return;
  if (p->dtype == SRMAP_F32) preload_sb<float>(z->S, z->B, z->regk, z->regr);
  else preload_sb<double>(z->S, z->B, z->regk, z->regr);
}

template <typename T, int S, int B>
static int dispatch_z(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g,
                      const T* wts, const ZPlan& z, int regk, int regr, double* partials, int* nb, hipStream_t st,
                      const T* dv, double* pgd, MFin mfin, bool ring_ahead) {
  if (regk == 1) return launch_z<T, S, B, 1, 0>(p, geo, obs_c0, terms, x, g, wts, z, partials, nb, st, dv, pgd, mfin, ring_ahead);
  if (regk == 2 && regr == 1) return launch_z<T, S, B, 2, 1>(p, geo, obs_c0, terms, x, g, wts, z, partials, nb, st, dv, pgd, mfin, ring_ahead);
  if (regk == 2 && regr == 2) return launch_z<T, S, B, 2, 2>(p, geo, obs_c0, terms, x, g, wts, z, partials, nb, st, dv, pgd, mfin, ring_ahead);
  if (regk == 2 && regr == 3) return launch_z<T, S, B, 2, 3>(p, geo, obs_c0, terms, x, g, wts, z, partials, nb, st, dv, pgd, mfin, ring_ahead);
  return launch_z<T, S, B, 0, 0>(p, geo, obs_c0, terms, x, g, wts, z, partials, nb, st, dv, pgd, mfin, ring_ahead);
}

template <typename T>
int launch_eval_ztile(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g,
                      double* partials, int* nblocks, hipStream_t st) {
  const ZPlan* zp = static_cast<const ZPlan*>(p->zplan);
  if (!zp) return set_error(p->ctx, SRMAP_EUNSUPPORTED, "no tile plan");
  const ZPlan& z = *zp;
  const size_t N = (size_t)geo.W * geo.H;
  const bool want_reg = (terms & SRMAP_TERM_REG) != 0;
  int regk = 0, regr = 0;
  const T* wts = nullptr;
  if (want_reg && z.regk != 0) {
    regk = z.regk; regr = z.regr;
    const RegSpec& rs = p->reg[z.reg_index];
    wts = rs.weights ? (const T*)rs.weights + (size_t)obs_c0 * N : nullptr;
  }
  unsigned zterms = terms & SRMAP_TERM_DATA;
  if (regk) zterms |= SRMAP_TERM_REG;
  int rc = SRMAP_OK, nb = 0;
  const int S = geo.s, B = geo.b;
  // g.d with the gradient (solver line search): only when this launch produces the WHOLE gradient and the single
  // finish launch reduces it -- no further regulariser kernels, few enough partials
  bool more_regs = false;
  if (want_reg)
    for (int r = 0; r < p->nreg; ++r)
      if (!(regk && r == z.reg_index) && p->reg[r].lambda > 0.0) more_regs = true;
  const size_t est_parts = ztile_est_partials(&z, geo.w, geo.H, geo.C);
  const bool sp_data = z.subpix && (terms & SRMAP_TERM_DATA);
  const bool with_d = p->eval_dvec != nullptr && g != nullptr && ztile_gd_instance_ok(p, z, geo.w, geo.H, geo.C, terms);
  int nfwd = 0;
  // the exact ring pass AHEAD of the tile kernel (k_gather_ring into the plan's buffer; the edge tiles add the values as
  // they store g, so g.d and the cost finish inside the tile launch) wherever the ring kernel applies; else behind it,
  // adding to g (then no g.d from the launch: ztile_gd_instance_ok).  Ring blocks at the front of the tile launch's own
  // grid (one launch less) were built and measured: every ring workgroup holds a tile slot (73 KB of LDS) for its 6 - 9 us of
  // latency, 450 of the 512 slots of the first generation -- 82.9 us against 82.5 us, and slower inside a solve.
  const bool ring_ahead = sp_data && g != nullptr && z.d_ringbuf != nullptr;
  if (sp_data) {
    // sub-pixel shifts: exact residuals (and the data cost) from the direct forward kernel, then the tile kernel
    // gathers them with the 4-tap tables; the pixels within Dr of the edge are evaluated exactly by the ring pass
    if (!p->d_resid) SRMAP_HIP(p->ctx, hipMalloc(&p->d_resid, p->lr_count() * sizeof(T)));
    SpFold sf;
    if (p->eval_fold_xk != nullptr) {
      if (!(with_d && z.spf.ok && z.spf.can_fold))
        return set_error(p->ctx, SRMAP_EINVAL, "internal: a folded trial point needs the forward tile kernel and the g.d instance (ztile_can_fold)");
      sf.xk = p->eval_fold_xk; sf.dvec = p->eval_dvec; sf.stp = p->eval_fold_stp; sf.norms = p->eval_fold_norms;
    }
    if (z.spf.ok)
      rc = launch_forward_sp<T>(p, geo, z.spf, x, (const T*)p->d_obs, p->geo.C, obs_c0, (T*)p->d_resid, partials, &nfwd, st, sf);
    else
      rc = launch_forward_direct<T>(p, geo, x, (const T*)p->d_obs, p->geo.C, obs_c0, (T*)p->d_resid, 0, geo.K, partials, &nfwd, st);
    if (rc) return rc;
    partials += nfwd;
    if (ring_ahead) {
      rc = launch_gather_direct<T>(p, geo, (const T*)p->d_resid, g, 0, geo.K, 2.0 * geo.s * geo.s, true, st, z.Dr, (T*)z.d_ringbuf);
      if (rc) return rc;
    }
  }
  if (p->eval_fold_xk != nullptr && !(with_d && p->ov_hook == nullptr))
    return set_error(p->ctx, SRMAP_EINVAL, "internal: a folded trial point needs the tile kernel's g.d instance (ztile_can_fold)");
  const T* dv = with_d ? (const T*)p->eval_dvec : nullptr;
  double* pgd = with_d ? p->d_partials + p->partials_cap / 2 : nullptr;
  p->gd_valid = false;
  p->eval_published = false;
  // tiles: the cost reduction inside the kernel (no finish launch) when no in-image pixel of the border frame needs a
  // correction, no further regulariser kernel follows and the granules suffice
  // Sub-pixel plan: the forward kernel's data-cost partials (plain doubles, complete before the tile kernel starts) are
  // added by the same in-kernel finish; the ring pass touches g only.
  MFin mfin;
  mfin.on = !more_regs && p->ov_hook == nullptr && z.d_mpart != nullptr && est_parts <= z.mpart_cap &&
            (z.n_ring == 0 || (z.ring.rg[0] == 0 && z.ring.rg[1] == 0)) && (size_t)nfwd <= kMaxFusedPartials;
  mfin.publish = mfin.on && with_d && p->eval_pub != nullptr;
  mfin.xpart = (mfin.on && sp_data) ? partials - nfwd : nullptr;
  mfin.n_xpart = (mfin.on && sp_data) ? nfwd : 0;
  auto tiles = [&]() {
#ifdef SRMAP_ZT_ONLY_CFG2
    if (sizeof(T) == sizeof(SRMAP_ZT_ONLY_T) && S == 4 && B == SRMAP_ZT_ONLY_B && regk == 2 && regr == 3)
      return launch_z<SRMAP_ZT_ONLY_T, 4, SRMAP_ZT_ONLY_B, 2, 3>(p, geo, obs_c0, zterms, (const SRMAP_ZT_ONLY_T*)x, (SRMAP_ZT_ONLY_T*)g,
                                                   (const SRMAP_ZT_ONLY_T*)wts, z, partials, &nb, st, (const SRMAP_ZT_ONLY_T*)dv, pgd,
                                                   mfin, ring_ahead);
    return set_error(p->ctx, SRMAP_EUNSUPPORTED, "measurement build: cfg2 instance only");
#else
    if (S == 2 && B == 1) return dispatch_z<T, 2, 1>(p, geo, obs_c0, zterms, x, g, wts, z, regk, regr, partials, &nb, st, dv, pgd, mfin, ring_ahead);
    if (S == 2 && B == 3) return dispatch_z<T, 2, 3>(p, geo, obs_c0, zterms, x, g, wts, z, regk, regr, partials, &nb, st, dv, pgd, mfin, ring_ahead);
    if (S == 3 && B == 1) return dispatch_z<T, 3, 1>(p, geo, obs_c0, zterms, x, g, wts, z, regk, regr, partials, &nb, st, dv, pgd, mfin, ring_ahead);
    if (S == 3 && B == 3) return dispatch_z<T, 3, 3>(p, geo, obs_c0, zterms, x, g, wts, z, regk, regr, partials, &nb, st, dv, pgd, mfin, ring_ahead);
    if (S == 4 && B == 1) return dispatch_z<T, 4, 1>(p, geo, obs_c0, zterms, x, g, wts, z, regk, regr, partials, &nb, st, dv, pgd, mfin, ring_ahead);
    if (S == 4 && B == 3) return dispatch_z<T, 4, 3>(p, geo, obs_c0, zterms, x, g, wts, z, regk, regr, partials, &nb, st, dv, pgd, mfin, ring_ahead);
    return set_error(p->ctx, SRMAP_EUNSUPPORTED, "no tile kernel for scale %d blur %d", S, B);
#endif
  };
  rc = tiles();
  if (rc) return rc;
  if (sp_data) {
    partials -= nfwd;
    nb += nfwd;
    if (g != nullptr && !ring_ahead) {
      rc = launch_gather_direct<T>(p, geo, (const T*)p->d_resid, g, 0, geo.K, 2.0 * geo.s * geo.s, true, st, z.Dr);
      if (rc) return rc;
    }
  }
  int total = nb;
  // remaining regularisers (3-D TV, a second regulariser, BTV range > 3): direct kernels, accumulating into g
  if (want_reg) {
    for (int r = 0; r < p->nreg; ++r) {
      if (regk && r == z.reg_index) continue;
      const RegSpec& rs = p->reg[r];
      if (rs.lambda <= 0.0) continue;
      const bool onfly = rs.kind != SRMAP_REG_BTV;
      if (!onfly) {
        if (!p->d_regvals) SRMAP_HIP(p->ctx, hipMalloc(&p->d_regvals, p->hr_count() * sizeof(T)));
        rc = launch_reg_values<T>(p, geo, rs, x, (T*)p->d_regvals, st);
        if (rc) return rc;
      }
      const T* w2 = rs.weights ? (const T*)rs.weights + (size_t)obs_c0 * N : nullptr;
      int nb2 = 0;
      rc = launch_reg_gradient_direct<T>(p, geo, rs, x, w2, rs.lambda, onfly ? nullptr : (const T*)p->d_regvals, g, true,
                                         partials + total, &nb2, st);
      if (rc) return rc;
      total += nb2;
    }
  }
  if (mfin.on) {  // reduced by the last workgroup of the tile kernel
    p->gd_valid = with_d;  // d_cost[1] = g.d
    p->eval_published = mfin.publish;
    *nblocks = 0;
    return SRMAP_OK;
  }
  // finish: border corrections of g + the fixed-order cost reduction, one launch
  const bool corr_on = (terms & SRMAP_TERM_DATA) && z.n_ring > 0 && g != nullptr;
  if ((size_t)total <= kMaxFusedPartials) {
    const int nring = corr_on ? z.n_ring : 0;
    const unsigned nb_f = 1u + (unsigned)((nring + 255) / 256);
    hipLaunchKernelGGL(k_finish_eval<T>, dim3(nb_f), dim3(256), 0, st, corr_on ? g : (T*)nullptr, (const T*)z.d_corr,
                       z.n_ring, geo.W, geo.H, z.ring, geo.C, (const double*)partials, total, p->d_cost, (const double*)pgd,
                       with_d ? p->eval_pub : (double*)nullptr, p->eval_pub_tag_slot, p->eval_pub_tag);
    SRMAP_HIP(p->ctx, hipGetLastError());
    p->gd_valid = with_d;  // d_cost[1] = g.d
    p->eval_published = with_d && p->eval_pub != nullptr;
    *nblocks = 0;  // total already in d_cost[0]
    return SRMAP_OK;
  }
  // many partials (multi-channel problems): corrections here, two-stage reduction by the caller
  if (corr_on) {
    hipLaunchKernelGGL(k_finish_eval<T>, dim3(1u + (unsigned)((z.n_ring + 255) / 256)), dim3(256), 0, st, g, (const T*)z.d_corr,
                       z.n_ring, geo.W, geo.H, z.ring, geo.C, (const double*)partials, 0, p->d_cost + 1, (const double*)nullptr,
                       (double*)nullptr, (double*)nullptr, 0.0);
    SRMAP_HIP(p->ctx, hipGetLastError());
  }
  *nblocks = total;
  return SRMAP_OK;
}

template int launch_eval_ztile<float>(srmap_problem*, const Geometry&, int, unsigned, const float*, float*,
                                      double*, int*, hipStream_t);
template int launch_eval_ztile<double>(srmap_problem*, const Geometry&, int, unsigned, const double*,
                                       double*, double*, int*, hipStream_t);

}  // namespace srmap
