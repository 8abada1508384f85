// kernels_tiled.hip -- the hot path: ONE fused kernel per MAP gradient
// iteration (ObjectiveFunction::ComputeAllTerms, objective_function.cpp:5-20)
// for the common geometry: integer motion shifts, HR = LR * S, S in {2,3,4},
// blur size B in {1,3}, and at most one regulariser handled in-kernel (2-D TV or
// BTV with range <= 3); anything else is evaluated by kernels_direct.hip.
//
// Design:
//   * Row-major ownership: a workgroup of TH waves (TH = 8 HR rows, 9 for
//     S = 3) owns a tile of TH x (64*S) HR pixels; wave w owns HR row w, lane l
//     owns the S pixels of LR cell column l in that row.  x / IRLS weights are
//     read and g is written as one S-element vector per thread (a wave touches
//     one contiguous 64*S-element row segment): every compulsory byte crosses
//     HBM once, x halo re-reads hit L2.  Row phase, the LR rows a pixel row
//     receives from and all row predicates are wave-uniform (SGPR).
//   * The x tile (+halo) is staged in LDS in POLYPHASE layout
//     xs[row][column phase (c mod S)][cell]: the decimated forward stencil
//     (stride-S access) and the per-pixel regulariser windows are unit-stride
//     across lanes -> no LDS bank conflicts, every tap offset an immediate.
//   * Per round of NW frames (one per wave): the wave sweeps the LR region the
//     tile's gather needs, one LR row per iteration of a rolled loop, and
//     computes r_k = A_k x - y_k (warp -> blur -> decimate fused,
//     objective_data_term.cpp:27-50) into LDS, stored frame-aligned; then every
//     thread gathers sum_k M_k^T B^T D^T r_k for its S pixels
//     (image_model.cpp:93-101) from the same LDS offsets for every frame, with
//     wave-uniform weights (1-D blur taps or 0) tabulated by the host: no
//     branches.  The forward stencil's column phase (ox mod S) selects one of S
//     code paths by a wave-uniform switch.
//   * The regulariser (tv_regularizer.cpp:135-227 / btv_regularizer.cpp:93-170,
//     bug-compatible) runs on the same x tile: pass 1 computes r, the self term
//     and 2*lambda*w*r for the tile plus an up/left halo strip into LDS, pass 2
//     adds the neighbour terms.
//   * Cost: fp64 wave-shuffle + workgroup reduction -> partials, summed in index
//     order (deterministic): by the last workgroup to arrive (agent-scope
//     ticket) on grids of about one generation of workgroups, by a separate
//     1-block launch on larger grids.
// No MFMA: this is a stencil/gather path.
#include "tiled_device.hpp"

namespace srmap {

namespace {

template <typename T, int B, int NP>
struct FusedArgs {
  const T* x;
  const T* y;
  const T* w;       // IRLS weights or nullptr
  T* g;             // nullptr = cost only
  double* partials;
  unsigned* counter;   // arrival counter for the in-kernel final reduction (nullptr = off)
  double* cost_out;    // receives the total when counter != nullptr
  unsigned long long* dbg;  // phase timeline (profiling aid, nullptr normally)
  const FrameInfo* frames;
  const int* gb;        // [K][S]   rs offset of the first LR tap of (frame, row phase)
  const T* wr;          // [K][S][2] 1-D blur weights of the (up to two) LR rows a pixel row receives from
  const T* wc;          // [K][S][2] the same per pixel column phase
  int W, H, wl, hl, K;
  int obs_C, obs_c0;
  int hu, hlc;          // x tile origin = (R0 - hu, cell CJ0 - hlc)
  int xrows, xcells;    // loaded extent
  int i0, j0;           // LR region origin relative to the tile's first cell (<= 0)
  int lrh, lrw;         // LR region extent
  int margin;           // tiles closer than this to the image edge take the border path
  int cr0, cr1;         // HR rows whose cost terms are counted (row-band sharding; default 0, H)
  int banded;           // tile rows on blockIdx.x, dealt to the 8 XCDs in contiguous bands (see the kernel)
  int terms;            // SRMAP_TERM_* (| ablation bits << 8, profiling only)
  T blur[B * B];        // k * k^T (blur_module.cpp:20-22)
  T lambda;
  T powtab[NP];         // BTV alpha^(i+j)
};

// Profiling aids (phase stamps, stop-after-stage, ablation bits) exist only in
// the -DSRMAP_PROFILING build (lib/libsrmap_prof.so, see tools/): in the product
// kernel they cost registers (the f32 instance spills with them compiled in).
#ifdef SRMAP_PROFILING
constexpr bool kProf = true;
#else
constexpr bool kProf = false;
#endif
#define SRMAP_STAMP(i)                                                                     \
  do {                                                                                     \
    if (kProf && A.dbg != nullptr && threadIdx.x == STAMP_TID)                             \
      A.dbg[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (i)] = \
          __builtin_amdgcn_s_memtime();                                                    \
    if (kProf && (A.terms >> 16) == (i) + 1) return; /* stop after stage i (uniform) */    \
  } while (0)

template <typename T, int S, int B, int REGK, int R>
__global__ __launch_bounds__((TileCfg<T, S>::NT), (sizeof(T) == 4 ? 6 : 4)) void k_eval_fused(
    FusedArgs<T, B, (REGK == 2 ? 2 * R + 1 : 1)> A) {
  using C = TileCfg<T, S>;
  constexpr int NP = (REGK == 2 ? 2 * R + 1 : 1);
  constexpr int RU = (REGK == 2) ? R - 1 : (REGK == 1 ? 1 : 0);
  constexpr int CR_ELEMS = (C::TH + RU) * C::CRROW;
  constexpr int SCRATCH_ELEMS = cmax_(C::RS_ELEMS, CR_ELEMS);
  __shared__ T xs[C::XS_ELEMS];
  __shared__ T scratch[SCRATCH_ELEMS];  // residuals during the data term, then 2*lambda*w*r
  __shared__ T gtw[kTabFrames * 4 * S];  // per frame: wr[S][2] then wc[S][2]
  __shared__ double red[2][C::NW];
  T* rs = scratch;
  T* cr = scratch;

  constexpr int STAMP_TID = 64 * (TileCfg<T, S>::NW - 1);  // last wave, lane 0
  SRMAP_STAMP(0);
  if (kProf && A.dbg != nullptr && threadIdx.x == STAMP_TID)  // HW_ID (CU / SE / XCC of this block) for the schedule plot
    A.dbg[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + 15] =
        ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index = HR row of the tile (SGPR)
  // profiling aid (bit 0x4000): every workgroup computes tile (4, 4) -- same instruction stream, all loads hit
  // the caches -- but writes its own tile: isolates how much of the time is exposed memory latency/bandwidth
  const bool same_tile = kProf && (A.terms & 0x4000) != 0;
  // XCD-aware tile order: workgroups are dealt to the 8 XCDs round robin in launch order, and each XCD has
  // its own L2.  With the tile ROW on blockIdx.x and launch index n -> row band (n mod 8), the workgroups
  // an XCD runs at the same time are vertical neighbours: the 12 halo rows of a 20-row x tile, the LR rows
  // of the observations and the halo IRLS weights they share are served by that XCD's L2 (cfg2: -4 %).
  // The map is a bijection for any row count: band b holds rows [b*q + min(b, r), ...), q = rows / 8, r = rows % 8.
  int tby = blockIdx.y;
  int tbx = blockIdx.x;
  if (A.banded) {
    const int n = blockIdx.x, q = gridDim.x >> 3, rem = gridDim.x & 7, bnd = n & 7;
    tby = bnd * q + (bnd < rem ? bnd : rem) + (n >> 3);
    tbx = blockIdx.y;
  }
  const int CI0 = (same_tile ? 4 : tby) * C::CH, CJ0 = (same_tile ? 4 : tbx) * C::CW;
  const int R0 = CI0 * S, C0 = CJ0 * S;
  const int ch = blockIdx.z;
  const size_t N = (size_t)A.W * A.H;
  const size_t nl = (size_t)A.wl * A.hl;
  const T* xplane = A.x + (size_t)ch * N;
  const bool border = (R0 < A.margin) || (R0 + C::TH + A.margin > A.H) || (C0 < A.margin) ||
                      (C0 + C::TW + A.margin > A.W);
  const int pr = wv % S, lci = wv / S;  // row phase and cell row of this wave (uniform)
  const int gr = R0 + wv;               // global HR row of this thread
  const int gc0 = C0 + S * lane;        // first global HR column of this thread
  const bool want_data = (A.terms & SRMAP_TERM_DATA) != 0;
  const bool want_reg = REGK != 0 && (A.terms & SRMAP_TERM_REG) != 0;

  // ---------------- prefetch: every global load whose address is known now ----------------
  // HBM/L2 latency (~1 us) is as long as a whole phase of this kernel, so loads
  // are issued as early as possible and consumed phases later.
  // ---------------- Phase A: x tile (+halo) -> LDS, polyphase ----------------
  // wave w stages rows w, w + NW, ...; lane = cell (0..63), the few cells beyond
  // 64 by the first lanes.  All loads are issued before the first LDS write.
  constexpr int ARI = (C::XR + C::NW - 1) / C::NW;  // row iterations per wave
  const bool skipA = kProf && (A.terms & 0x400) != 0;
  const int extra = A.xcells - 64;                  // 0..kMaxHaloCells
  T va[ARI][S], vb[ARI][S];
#pragma unroll
    for (int it = 0; it < ARI; ++it) {
      const int row = wv + it * C::NW;
      const int grr = R0 - A.hu + row;
      const bool row_in = !skipA && row < A.xrows && (unsigned)grr < (unsigned)A.H;  // uniform
      const int gca = CJ0 - A.hlc + lane, gcb = gca + 64;
      const bool ina = row_in && (unsigned)gca < (unsigned)A.wl;
      const bool inb = row_in && lane < extra && (unsigned)gcb < (unsigned)A.wl;
      const T* sa = xplane + (ina ? (size_t)grr * A.W + (size_t)gca * S : (size_t)0);
      const T* sb = xplane + (inb ? (size_t)grr * A.W + (size_t)gcb * S : (size_t)0);
#pragma unroll
      for (int pc = 0; pc < S; ++pc) va[it][pc] = sa[pc];
#pragma unroll
      for (int pc = 0; pc < S; ++pc) vb[it][pc] = sb[pc];
#pragma unroll
      for (int pc = 0; pc < S; ++pc) { va[it][pc] = ina ? va[it][pc] : T(0); vb[it][pc] = inb ? vb[it][pc] : T(0); }
    }

  // ---- the other prefetches go out behind the x tile (loads return in order) ----
  const int gi0 = CI0 + A.i0, gj0 = CJ0 + A.j0;  // LR region origin
  const LaneTail tl = tail_lane(lane, A.lrh, A.lrw);
  ObsPrefetch<T> op = {T(0), T(0), T(0)};
  if (want_data && wv < A.K && !(kProf && (A.terms & 0x1000)))
    op = prefetch_obs<T>(A, A.y + ((size_t)wv * A.obs_C + ch + A.obs_c0) * nl, lane, tl, gi0, gj0);
  // only tiles whose LR region contains LR row 0 or column 0 need blur-tap masks
  const bool edge_tl = (gi0 <= 0) || (gj0 <= 0);
  // gather tables: NTAB elements per thread into registers now, into LDS after the x tile
  constexpr int NTAB = (kTabFrames * 4 * S + C::NT - 1) / C::NT;
  const int kt = A.K < kTabFrames ? A.K : kTabFrames;
  const bool tab_on = want_data && A.g != nullptr;
  T tabw[NTAB];
#pragma unroll
  for (int q = 0; q < NTAB; ++q) {
    tabw[q] = T(0);
    const int e = tid + q * C::NT;
    if (tab_on && q * C::NT < kt * 4 * S) {  // uniform
      const int tk = e / (4 * S), ti = e - tk * 4 * S;
      const bool okw = e < kt * 4 * S;
      tabw[q] = ti < 2 * S ? A.wr[okw ? tk * 2 * S + ti : 0] : A.wc[okw ? tk * 2 * S + ti - 2 * S : 0];
    }
  }
  T wreg[S];  // IRLS weights of this thread's pixels
  using HG = HaloGeom<T, S, REGK, R>;
  T whalo[HG::NIT];  // ... and of its halo-strip pixel(s)
  if (want_reg) {
    const T* wplane = A.w ? A.w + (size_t)ch * N : nullptr;
    const bool in = wplane != nullptr && gr < A.H && gc0 < A.W;
#pragma unroll
    for (int pc = 0; pc < S; ++pc) wreg[pc] = in ? wplane[(size_t)gr * A.W + gc0 + pc] : T(1);
#pragma unroll
    for (int it = 0; it < HG::NIT; ++it) {
      int hrow = 0, hcol = 0;
      const int h = tid + it * C::NT;
      halo_pixel<T, S, REGK, R>(h < HG::NH ? h : 0, hrow, hcol);
      const int hgr = R0 + hrow, hgc = C0 + hcol;
      const bool hin = wplane != nullptr && A.g != nullptr && h < HG::NH && (unsigned)hgr < (unsigned)A.H &&
                       (unsigned)hgc < (unsigned)A.W;
      whalo[it] = wplane != nullptr ? wplane[hin ? (size_t)hgr * A.W + hgc : (size_t)0] : T(1);
      if (!hin && wplane != nullptr) whalo[it] = T(1);
    }
  }

  SRMAP_STAMP(1);
  // stage 2 of Phase A: polyphase scatter into LDS
  {
#pragma unroll
    for (int it = 0; it < ARI; ++it) {
      const int row = wv + it * C::NW;
      if (row < A.xrows) {  // uniform
#pragma unroll
        for (int pc = 0; pc < S; ++pc) xs[row * C::XROW + pc * C::XPLANE + lane] = va[it][pc];
        if (lane < extra) {
#pragma unroll
          for (int pc = 0; pc < S; ++pc) xs[row * C::XROW + pc * C::XPLANE + 64 + lane] = vb[it][pc];
        }
      }
    }
  }
  if (tab_on) {
#pragma unroll
    for (int q = 0; q < NTAB; ++q)
      if (tid + q * C::NT < kt * 4 * S) gtw[tid + q * C::NT] = tabw[q];
  }
  SRMAP_STAMP(2);
  __syncthreads();
  SRMAP_STAMP(3);

  T acc[S];
#pragma unroll
  for (int j = 0; j < S; ++j) acc[j] = T(0);
  double cost_data = 0.0, cost_reg = 0.0;

  if (want_data) {
    for (int k0 = 0; k0 < A.K; k0 += C::FR) {
      // ---------------- Phase B: residuals of frame k0 + wave ----------------
      const int k = k0 + wv;  // wave-uniform
      if (k < A.K && !(kProf && (A.terms & 0x100))) {
        const int soff = A.frames[k].frow * C::XROW + A.frames[k].fcell;  // scalar loads
        const int fxm = A.frames[k].fxm;
        const int sy = A.frames[k].sy, sx = A.frames[k].sx;
        T* rsk = rs + wv * (C::GRH * C::GRW);
        const T* yk = A.y + ((size_t)k * A.obs_C + ch + A.obs_c0) * nl;
        if (edge_tl) residual_switch<T, S, B, true>(A, xs, rsk, yk, op, lane, tl, gi0, gj0, CI0, CJ0, soff, sy, sx, fxm, cost_data);
        else residual_switch<T, S, B, false>(A, xs, rsk, yk, op, lane, tl, gi0, gj0, CI0, CJ0, soff, sy, sx, fxm, cost_data);
      }
      if (k0 == 0) SRMAP_STAMP(4);
      // observations of the next round: in flight during the gather
      if (k + C::FR < A.K)
        op = prefetch_obs<T>(A, A.y + ((size_t)(k + C::FR) * A.obs_C + ch + A.obs_c0) * nl, lane, tl, gi0, gj0);
      __syncthreads();
      if (k0 == 0) SRMAP_STAMP(5);
      // ---------------- Phase C: gather into this thread's S pixels ----------------
      if (A.g != nullptr && !(kProf && (A.terms & 0x200))) {
        const int kc = (A.K - k0) < C::FR ? (A.K - k0) : C::FR;
        // every frame's 2 x 2 LR patch of this thread sits at the same LDS
        // offsets (frame-aligned storage), the weights are LDS broadcasts: no
        // dependent address chain, the unrolled loop keeps 4 frames in flight
        const T* rs_row = rs + lci * C::GRW + lane;
        const T* gwr = gtw + 2 * pr;
        if (!border) {
#pragma unroll 4
          for (int kk = 0; kk < kc; ++kk) {
            const T* gw = gwr + (k0 + kk) * 4 * S;
            T wcv[2 * S];
#pragma unroll
            for (int i = 0; i < 2 * S; ++i) wcv[i] = gw[2 * S - 2 * pr + i];
            gather_frame<T, S, false>(acc, rs_row + kk * (C::GRH * C::GRW), gw[0], gw[1], wcv, 0xffffffffu);
          }
        } else {
          for (int kk = 0; kk < kc; ++kk) {
            const int k = k0 + kk;
            const int toy = A.frames[k].toy, tox = A.frames[k].tox;
            if ((unsigned)(gr + toy) >= (unsigned)A.H) continue;  // p' row outside the image (uniform)
            const T* gw = gwr + k * 4 * S;
            T wcv[2 * S];
#pragma unroll
            for (int i = 0; i < 2 * S; ++i) wcv[i] = gw[2 * S - 2 * pr + i];
            // column masks only when this frame's shift moves some pixel of the tile out of the image (uniform)
            if (C0 + tox >= 0 && C0 + C::TW - 1 + tox < A.W) {
              gather_frame<T, S, false>(acc, rs_row + kk * (C::GRH * C::GRW), gw[0], gw[1], wcv, 0xffffffffu);
            } else {
              unsigned cmask = 0;
#pragma unroll
              for (int pc = 0; pc < S; ++pc) cmask |= ((unsigned)(gc0 + pc + tox) < (unsigned)A.W ? 1u : 0u) << pc;
              gather_frame<T, S, true>(acc, rs_row + kk * (C::GRH * C::GRW), gw[0], gw[1], wcv, cmask);
            }
          }
        }
      }
      if (k0 == 0) SRMAP_STAMP(6);
      __syncthreads();
      if (k0 == 0) SRMAP_STAMP(7);
    }
    SRMAP_STAMP(8);
    const T sc = (T)(2 * S * S);  // g += 2 * (s*s block sum) (objective_data_term.cpp:55-71)
#pragma unroll
    for (int j = 0; j < S; ++j) acc[j] *= sc;
  }

  // ---------------- Phase D: regulariser ----------------
  if (want_reg) {
    const T* wplane = A.w ? A.w + (size_t)ch * N : nullptr;
    const int xrow = A.hu + wv, xcell = A.hlc + lane;
    // pass 1 looks right / down only: tap masks are needed by tiles at the right or bottom image edge alone
    constexpr int WIN1 = (REGK == 2) ? R : 1;
    const bool reg_border = (R0 + C::TH + WIN1 > A.H) || (C0 + C::TW + WIN1 > A.W);
    if (reg_border)
      reg_pass1<T, S, REGK, R, NP, true>(acc, cost_reg, xs, cr, wreg, xrow, xcell, wv + RU, lane + 1, gr, gc0, A.W,
                                         A.H, A.lambda, A.powtab, gr >= A.cr0 && gr < A.cr1);
    else
      reg_pass1<T, S, REGK, R, NP, false>(acc, cost_reg, xs, cr, wreg, xrow, xcell, wv + RU, lane + 1, gr, gc0, A.W,
                                          A.H, A.lambda, A.powtab, gr >= A.cr0 && gr < A.cr1);
    SRMAP_STAMP(9);
    if (A.g != nullptr)  // halo strips are needed by pass 2 only
      reg_halo<T, S, REGK, R, NP>(xs, cr, whalo, tid, A.hu, A.hlc, R0, C0, A.W, A.H, A.lambda, A.powtab);
    SRMAP_STAMP(10);
    __syncthreads();
    SRMAP_STAMP(11);
    if (A.g != nullptr)
      reg_pass2<T, S, REGK, R, NP>(acc, xs, cr, xrow, xcell, wv + RU, lane + 1, A.powtab);
  }

  SRMAP_STAMP(12);
  // ---------------- write g: one S-element vector per thread, a wave = one row segment ----------------
  if (A.g != nullptr && gr < A.H && gc0 < A.W && !(kProf && (A.terms & 0x2000))) {
    T* dst = A.g + (size_t)ch * N + (size_t)gr * A.W + gc0;
    if (same_tile) dst += ((size_t)tby - 4) * C::TH * A.W + ((size_t)tbx - 4) * C::TW;
#pragma unroll
    for (int pc = 0; pc < S; ++pc) __builtin_nontemporal_store(acc[pc], &dst[pc]);  // written once, not re-read here
  }

  SRMAP_STAMP(13);
  // ---------------- cost partials ----------------
  {
    const double sd = wave_sum_d(cost_data);
    const double sr = wave_sum_d(cost_reg);
    if (lane == 0) { red[0][wv] = sd; red[1][wv] = sr; }
    __syncthreads();
    // Only wave 0 continues: the other waves retire here, so nothing waits for
    // the device-scope round trips below.
    if (wv == 0) {
      const unsigned nblocks = gridDim.x * gridDim.y * gridDim.z;
      double d = 0.0, r = 0.0;
#pragma unroll
      for (int i = 0; i < C::NW; ++i) { d += red[0][i]; r += red[1][i]; }
      const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      const double part = (double)(S * S) * d + r;
      if (A.counter == nullptr) {
        if (lane == 0) A.partials[b] = part;
      } else {
        // publish write-through (sc1), drain, then take a ticket: the last
        // arriver sums all partials in index order -> deterministic total
        // without a second launch (cdna guide, G16 "R1" form).  One counter
        // saturates at ~88 atomics/us, so arrivals are spread over kSubCounters
        // words (256 B apart) with a second-level counter on top.
        int last = 0;
        if (lane == 0) {
          __hip_atomic_store(&A.partials[b], part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const unsigned sub = (unsigned)b % kSubCounters;
          const unsigned expect = (nblocks - sub + kSubCounters - 1) / kSubCounters;  // blocks mapped to this word
          const unsigned nsub = nblocks < kSubCounters ? nblocks : kSubCounters;
          const unsigned t1 = __hip_atomic_fetch_add(A.counter + 64 * (1 + sub), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (t1 == expect - 1) {
            __hip_atomic_store(A.counter + 64 * (1 + sub), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned t2 = __hip_atomic_fetch_add(A.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (t2 == nsub - 1) ? 1 : 0;
          }
        }
        last = __builtin_amdgcn_readfirstlane(last);
        if (last) {  // this wave belongs to the last workgroup: fixed-order sum of all partials
          double v = 0.0;
          for (unsigned i = lane; i < nblocks; i += 64)
            v += __hip_atomic_load(&A.partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v = wave_sum_d(v);
          if (lane == 0) {
            A.cost_out[0] = v;
            __hip_atomic_store(A.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
          }
        }
      }
    }
  }
  SRMAP_STAMP(14);
}

// Decide whether the fused kernel covers the problem and derive the tile halos.
static HostPlan make_plan(const srmap_problem* p, int S, int CH, int CW) {
  HostPlan pl;
  const Geometry& g = p->geo;
  const int B = g.b, hb = g.hb, K = g.K;
  const int TH = CH * S;
  if (!p->maps_regular) return pl;
  if (B != 1 && B != 3) return pl;
  if (K > kTabFrames) return pl;  // gather weights of every frame are staged in LDS
  // integer shifts only
  std::vector<int> ox(K, 0), oy(K, 0), tx(K, 0), ty(K, 0);
  if (p->has_motion) {
    for (int k = 0; k < K; ++k) {
      if (p->fwd_warps[k].ntaps != 1 || p->bwd_warps[k].ntaps != 1) return pl;
      ox[k] = p->fwd_warps[k].ox; oy[k] = p->fwd_warps[k].oy;
      tx[k] = p->bwd_warps[k].ox; ty[k] = p->bwd_warps[k].oy;
      if (std::abs(tx[k]) > 30000 || std::abs(ty[k]) > 30000) return pl;
    }
  }
  // the one regulariser handled in-kernel (first TV / BTV with lambda > 0)
  for (int r = 0; r < p->nreg && pl.regk == 0; ++r) {
    const RegSpec& rs = p->reg[r];
    if (rs.lambda <= 0) continue;
    if (rs.kind == SRMAP_REG_TV) { pl.regk = 1; pl.reg_index = r; }
    else if (rs.kind == SRMAP_REG_BTV && rs.range >= 1 && rs.range <= 3 && rs.range - 1 <= S) {
      pl.regk = 2; pl.regr = rs.range; pl.reg_index = r;
    }
    break;  // only the first active regulariser may be fused (order of accumulation)
  }
  // LR region needed by the gather (+ the owned pixels)
  int dimin = 0, dimax = 0, djmin = 0, djmax = 0;
  int amax = 0;
  for (int k = 0; k < K; ++k) {
    dimin = std::min(dimin, -fdiv(-(ty[k] - hb), S));            // ceil((toy-hb)/S)
    dimax = std::max(dimax, fdiv(ty[k] + S - 1 + B - 1 - hb, S));
    dimax = std::max(dimax, fdiv(ty[k], S) + 1);   // the gather reads the whole 2 x 2 LR patch of a frame
    djmax = std::max(djmax, fdiv(tx[k], S) + 1);
    dimin = std::min(dimin, fdiv(ty[k], S));
    djmin = std::min(djmin, fdiv(tx[k], S));
    djmin = std::min(djmin, -fdiv(-(tx[k] - hb), S));
    djmax = std::max(djmax, fdiv(tx[k] + S - 1 + B - 1 - hb, S));
    amax = std::max(amax, std::max(std::abs(ty[k]), std::abs(tx[k])));
    amax = std::max(amax, std::max(std::abs(oy[k]), std::abs(ox[k])));
  }
  pl.i0 = dimin; pl.j0 = djmin;
  pl.lrh = CH + dimax - dimin; pl.lrw = CW + djmax - djmin;
  if (pl.lrh > CH + 3 || pl.lrw > CW + 3) return pl;
  // x rows/cols the forward model touches for those LR pixels (tile-relative)
  int rmin = 0, rmax = TH - 1, cmin = 0, cmax = CW * S - 1;
  for (int k = 0; k < K; ++k) {
    rmin = std::min(rmin, S * dimin - hb + oy[k]);
    rmax = std::max(rmax, S * (CH - 1 + dimax) + (B - 1 - hb) + oy[k]);
    cmin = std::min(cmin, S * djmin - hb + ox[k]);
    cmax = std::max(cmax, S * (CW - 1 + djmax) + (B - 1 - hb) + ox[k]);
  }
  if (pl.regk) {
    const int win = pl.regk == 2 ? pl.regr : 1;
    const int ru = pl.regk == 2 ? pl.regr - 1 : 1;
    rmin = std::min(rmin, -ru);                     // halo strip rows
    if (ru > 0) cmin = std::min(cmin, -S);          // halo strip columns live in one halo cell
    rmax = std::max(rmax, TH - 1 + win);
    cmax = std::max(cmax, CW * S - 1 + win);
  }
  pl.hu = -rmin;
  pl.hd = rmax - (TH - 1);
  pl.hlc = -fdiv(cmin, S);
  pl.hrc = fdiv(cmax, S) - (CW - 1);
  if (pl.hu + pl.hd > kMaxHaloRows || pl.hlc + pl.hrc > kMaxHaloCells) return pl;
  pl.margin = std::max(std::max(pl.hu, pl.hd), std::max(pl.hlc, pl.hrc) * S) + amax + hb + 4;
  pl.frames.resize(K);
  for (int k = 0; k < K; ++k) {
    FrameInfo& f = pl.frames[k];
    f.frow = S * pl.i0 + oy[k] + pl.hu - hb;
    f.fcell = pl.j0 + pl.hlc + fdiv(ox[k], S);
    f.fxm = pmod(ox[k], S);
    f.sy = fdiv(ty[k], S) - pl.i0;
    f.sx = fdiv(tx[k], S) - pl.j0;
    f.gym = pmod(ty[k], S);
    f.gxm = pmod(tx[k], S);
    f.toy = ty[k]; f.tox = tx[k];
  }
  pl.ok = true;
  return pl;
}

}  // namespace

// The plan is rebuilt whenever the regulariser list changes (cheap, host only)
// and cached per problem.
static std::vector<std::pair<const srmap_problem*, PlanCache>>& plan_table() {
  static std::vector<std::pair<const srmap_problem*, PlanCache>> t;
  return t;
}

PlanCache* tiled_find_plan(const srmap_problem* p) {
  for (auto& e : plan_table()) if (e.first == p) return &e.second;
  return nullptr;
}

void tiled_release(srmap_problem* p) {
  auto& t = plan_table();
  for (size_t i = 0; i < t.size(); ++i)
    if (t[i].first == p) {
      if (t[i].second.d_frames) (void)hipFree(t[i].second.d_frames);
      if (t[i].second.d_gb) (void)hipFree(t[i].second.d_gb);
      if (t[i].second.d_wr) (void)hipFree(t[i].second.d_wr);
      if (t[i].second.d_wc) (void)hipFree(t[i].second.d_wc);
      t.erase(t.begin() + i);
      return;
    }
}

bool tiled_plan(srmap_problem* p) {
  tiled_release(p);
  const int S = p->geo.s;
  HostPlan pl;
  if (S == 2) pl = make_plan(p, 2, TileCfg<float, 2>::CH, TileCfg<float, 2>::CW);
  else if (S == 3) pl = make_plan(p, 3, TileCfg<float, 3>::CH, TileCfg<float, 3>::CW);
  else if (S == 4) pl = make_plan(p, 4, TileCfg<float, 4>::CH, TileCfg<float, 4>::CW);
  if (!pl.ok) return false;
  PlanCache pc;
  pc.plan = pl;
  // gather tables: for (frame k, row phase pr) the first LR row a pixel row
  // receives from and the 1-D blur weights of that row and the next one; the
  // same per column phase.  R = phase + toy_mod + a - hb must be a multiple of S
  // (zero insertion); its quotient selects LR row 0 or 1 of the 2 x 2 patch.
  const int K = p->geo.K, B = p->geo.b, hb = p->geo.hb;
  const int LRW = TileCfg<float, 2>::LRW;  // CW + 3 for every S
  const std::vector<double>& k1 = p->blur1d;
  std::vector<int> gb((size_t)K * S);
  std::vector<double> wr((size_t)K * S * 2, 0.0), wc((size_t)K * S * 2, 0.0);
  bool patch_ok = true;
  for (int k = 0; k < K; ++k) {
    const FrameInfo& f = pl.frames[k];
    for (int ph = 0; ph < S; ++ph) {
      for (int a = 0; a < B; ++a) {
        const int R = ph + f.gym + a - hb;  // row of the zero-inserted image relative to S*(li base)
        if (pmod(R, S) != 0) continue;
        const int d = fdiv(R, S);
        if (d < 0 || d > 1) { patch_ok = false; continue; }
        wr[((size_t)k * S + ph) * 2 + d] += k1[a];
      }
      for (int e = 0; e < B; ++e) {
        const int Cc = ph + f.gxm + e - hb;
        if (pmod(Cc, S) != 0) continue;
        const int d = fdiv(Cc, S);
        if (d < 0 || d > 1) { patch_ok = false; continue; }
        wc[((size_t)k * S + ph) * 2 + d] += k1[e];
      }
      gb[(size_t)k * S + ph] = 0;
    }
  }
  if (!patch_ok) return false;
  auto up = [&](const std::vector<double>& v, void** d) {
    if (p->dtype == SRMAP_F32) {
      std::vector<float> t(v.begin(), v.end());
      return hipMalloc(d, t.size() * 4) == hipSuccess && hipMemcpy(*d, t.data(), t.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    }
    return hipMalloc(d, v.size() * 8) == hipSuccess && hipMemcpy(*d, v.data(), v.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
  };
  bool ok = hipMalloc((void**)&pc.d_frames, sizeof(FrameInfo) * pl.frames.size()) == hipSuccess &&
            hipMemcpy(pc.d_frames, pl.frames.data(), sizeof(FrameInfo) * pl.frames.size(), hipMemcpyHostToDevice) == hipSuccess &&
            hipMalloc((void**)&pc.d_gb, sizeof(int) * gb.size()) == hipSuccess &&
            hipMemcpy(pc.d_gb, gb.data(), sizeof(int) * gb.size(), hipMemcpyHostToDevice) == hipSuccess &&
            up(wr, &pc.d_wr) && up(wc, &pc.d_wc);
  if (!ok) {
    if (pc.d_frames) (void)hipFree(pc.d_frames);
    if (pc.d_gb) (void)hipFree(pc.d_gb);
    if (pc.d_wr) (void)hipFree(pc.d_wr);
    if (pc.d_wc) (void)hipFree(pc.d_wc);
    return false;
  }
  plan_table().push_back({p, pc});
  return true;
}

template <typename T, int S, int B, int REGK, int R>
static int launch_fused(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g,
                        const T* wts, const PlanCache& pc, double* partials, int* nblocks, bool final_reduce,
                        hipStream_t st) {
  using C = TileCfg<T, S>;
  constexpr int NP = (REGK == 2 ? 2 * R + 1 : 1);
  const HostPlan& pl = pc.plan;
  FusedArgs<T, B, NP> A;
  A.x = x; A.y = (const T*)p->d_obs; A.w = wts; A.g = g; A.partials = partials;
  A.counter = final_reduce ? p->d_counters : nullptr;  // zero-initialised, reset by the last workgroup
  A.cost_out = p->d_cost;
  A.frames = pc.d_frames;
  A.gb = pc.d_gb; A.wr = (const T*)pc.d_wr; A.wc = (const T*)pc.d_wc;
  A.W = geo.W; A.H = geo.H; A.wl = geo.w; A.hl = geo.h; A.K = geo.K;
  A.obs_C = p->geo.C; A.obs_c0 = obs_c0;
  A.hu = pl.hu; A.hlc = pl.hlc;
  A.xrows = C::TH + pl.hu + pl.hd;
  A.xcells = C::CW + pl.hlc + pl.hrc;
  A.i0 = pl.i0; A.j0 = pl.j0; A.lrh = pl.lrh; A.lrw = pl.lrw;
  A.margin = pl.margin;
  A.cr0 = geo.cr0; A.cr1 = geo.cr1;
  A.terms = (int)terms;
  if (kProf) {  // libsrmap_prof.so only
    if (const char* dbg = getenv("SRMAP_DEBUG_SKIP")) A.terms |= atoi(dbg) << 8;         // ablation bits
    if (const char* dbg = getenv("SRMAP_DEBUG_STOP")) A.terms |= (atoi(dbg) + 1) << 16;  // stop after stamp n
  }
  for (int i = 0; i < B * B; ++i) A.blur[i] = (T)p->blur2d[i];
  A.lambda = T(0);
  for (int i = 0; i < NP; ++i) A.powtab[i] = T(1);
  if (REGK != 0) {
    const RegSpec& rs = p->reg[pl.reg_index];
    A.lambda = (T)rs.lambda;
    if (REGK == 2) for (int i = 0; i < NP; ++i) A.powtab[i] = (T)rs.pow_table[i];
  }
  dim3 grid((geo.w + C::CW - 1) / C::CW, (geo.h + C::CH - 1) / C::CH, geo.C);
  A.banded = getenv("SRMAP_TILE_ORDER_NATURAL") ? 0 : 1;  // A/B aid
  if (A.banded) { const unsigned t = grid.x; grid.x = grid.y; grid.y = t; }
  A.dbg = nullptr;
  static int tl_calls = 0;
  const bool timeline = kProf && getenv("SRMAP_DEBUG_TIMELINE") != nullptr && ++tl_calls == 30;
  const size_t nb_ = (size_t)grid.x * grid.y * grid.z;
  if (timeline) { (void)hipMalloc((void**)&A.dbg, nb_ * 16 * 8); (void)hipMemset(A.dbg, 0, nb_ * 16 * 8); }
  static const int extra_lds = getenv("SRMAP_DEBUG_EXTRA_LDS") ? atoi(getenv("SRMAP_DEBUG_EXTRA_LDS")) : 0;  // occupancy probe
  hipLaunchKernelGGL((k_eval_fused<T, S, B, REGK, R>), grid, dim3(C::NT), extra_lds, st, A);
  if (timeline) {
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> h(nb_ * 16);
    (void)hipMemcpy(h.data(), A.dbg, nb_ * 16 * 8, hipMemcpyDeviceToHost);
    double sum[16] = {0}; unsigned long long t0 = ~0ull, t1 = 0;
    size_t nint = 0;
    for (size_t b = 0; b < nb_; ++b) {
      const size_t bx = b % grid.x, by = (b / grid.x) % grid.y;
      if (getenv("SRMAP_TIMELINE_INTERIOR") && (bx == 0 || bx + 1 >= grid.x || by < 3 || by + 3 >= grid.y)) continue;
      ++nint;
      for (int i = 1; i < 15; ++i) { unsigned long long prev = 0; for (int j = i - 1; j >= 0; --j) if (h[b * 16 + j]) { prev = h[b * 16 + j]; break; } if (h[b * 16 + i] && prev) sum[i] += (double)(h[b * 16 + i] - prev); }
      if (h[b * 16]) t0 = std::min(t0, h[b * 16]);
      for (int i = 0; i < 15; ++i) t1 = std::max(t1, h[b * 16 + i]);
    }
    fprintf(stderr, "[timeline] blocks %zu, span %llu ticks (s_memtime = shader clock); mean delta per stamp:", nb_, t1 - t0);
    for (int i = 1; i < 15; ++i) fprintf(stderr, " %d:%.0f", i, sum[i] / (nint ? nint : 1));
    fprintf(stderr, " (over %zu blocks)", nint);
    fprintf(stderr, "\n");
    if (const char* f = getenv("SRMAP_DEBUG_TIMELINE_FILE")) { FILE* fp = fopen(f, "wb"); if (fp) { fwrite(h.data(), 8, h.size(), fp); fclose(fp); } }
    (void)hipFree(A.dbg);
  }
  *nblocks = (int)(grid.x * grid.y * grid.z);
  SRMAP_HIP(p->ctx, hipGetLastError());
  return SRMAP_OK;
}

template <typename T, int S, int B>
static int dispatch_reg(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g,
                        const T* wts, const PlanCache& pc, int regk, int regr, double* partials, int* nb,
                        bool fr, hipStream_t st) {
  if (regk == 1) return launch_fused<T, S, B, 1, 0>(p, geo, obs_c0, terms, x, g, wts, pc, partials, nb, fr, st);
  if (regk == 2 && regr == 1) return launch_fused<T, S, B, 2, 1>(p, geo, obs_c0, terms, x, g, wts, pc, partials, nb, fr, st);
  if (regk == 2 && regr == 2) return launch_fused<T, S, B, 2, 2>(p, geo, obs_c0, terms, x, g, wts, pc, partials, nb, fr, st);
  if (regk == 2 && regr == 3) return launch_fused<T, S, B, 2, 3>(p, geo, obs_c0, terms, x, g, wts, pc, partials, nb, fr, st);
  return launch_fused<T, S, B, 0, 0>(p, geo, obs_c0, terms, x, g, wts, pc, partials, nb, fr, st);
}

template <typename T>
int launch_eval_tiled(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g,
                      double* partials, int* nblocks, hipStream_t st) {
  PlanCache* pc = tiled_find_plan(p);
  if (!pc) return set_error(p->ctx, SRMAP_EUNSUPPORTED, "no tile plan");
  const HostPlan& pl = pc->plan;
  const size_t N = (size_t)geo.W * geo.H;
  // which regularisers the fused kernel takes, which go to the direct kernels
  int regk = 0, regr = 0;
  const bool want_reg = (terms & SRMAP_TERM_REG) != 0;
  const T* wts = nullptr;
  if (want_reg && pl.regk != 0) {
    regk = pl.regk; regr = pl.regr;
    const RegSpec& rs = p->reg[pl.reg_index];
    wts = rs.weights ? (const T*)rs.weights + (size_t)obs_c0 * N : nullptr;
  }
  unsigned fused_terms = terms & SRMAP_TERM_DATA;
  if (regk) fused_terms |= SRMAP_TERM_REG;
  // the fused kernel finishes the cost reduction itself when it is the only
  // producer of partials
  bool extra = false;
  if (want_reg)
    for (int r = 0; r < p->nreg; ++r)
      if (!(regk && r == pl.reg_index) && p->reg[r].lambda > 0.0) extra = true;
  // ... and when the grid is about one generation of workgroups.  On larger grids
  // the ticket's device-scope round trip keeps wave 0 -- and with it the
  // workgroup's LDS -- resident for microseconds and delays the next workgroup
  // of that CU; a separate 1-block reduction launch is cheaper (measured at
  // cfg2: 107 -> 100 us f64, 75.5 -> 68.9 us f32).
  const long ntiles_est = (long)((geo.w + 63) / 64) * ((geo.h * geo.s + 7) / 8) * geo.C;
  bool fr = !extra && ntiles_est <= 2L * (p->ctx->num_cus > 0 ? p->ctx->num_cus : 256);
  if (const char* e = getenv("SRMAP_INKERNEL_REDUCE")) fr = !extra && atoi(e) != 0;  // profiling override
  int rc = SRMAP_OK, nb = 0;
  if (geo.s == 2 && geo.b == 1) rc = dispatch_reg<T, 2, 1>(p, geo, obs_c0, fused_terms, x, g, wts, *pc, regk, regr, partials, &nb, fr, st);
  else if (geo.s == 2 && geo.b == 3) rc = dispatch_reg<T, 2, 3>(p, geo, obs_c0, fused_terms, x, g, wts, *pc, regk, regr, partials, &nb, fr, st);
  else if (geo.s == 3 && geo.b == 1) rc = dispatch_reg<T, 3, 1>(p, geo, obs_c0, fused_terms, x, g, wts, *pc, regk, regr, partials, &nb, fr, st);
  else if (geo.s == 3 && geo.b == 3) rc = dispatch_reg<T, 3, 3>(p, geo, obs_c0, fused_terms, x, g, wts, *pc, regk, regr, partials, &nb, fr, st);
  else if (geo.s == 4 && geo.b == 1) rc = dispatch_reg<T, 4, 1>(p, geo, obs_c0, fused_terms, x, g, wts, *pc, regk, regr, partials, &nb, fr, st);
  else if (geo.s == 4 && geo.b == 3) rc = dispatch_reg<T, 4, 3>(p, geo, obs_c0, fused_terms, x, g, wts, *pc, regk, regr, partials, &nb, fr, st);
  else return set_error(p->ctx, SRMAP_EUNSUPPORTED, "no fused kernel for scale %d blur %d", geo.s, geo.b);
  if (rc) return rc;
  if (fr) { *nblocks = 0; return SRMAP_OK; }  // total already in d_cost[0]
  int total = nb;
  // remaining regularisers (TV3D, a second regulariser, BTV range > 3): direct kernels
  if (want_reg) {
    for (int r = 0; r < p->nreg; ++r) {
      if (regk && r == pl.reg_index) continue;
      const RegSpec& rs = p->reg[r];
      if (rs.lambda <= 0.0) continue;
      // TV / 3-D TV: one pass, values recomputed in the gradient kernel; BTV: values first
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
  *nblocks = total;
  return SRMAP_OK;
}

template int launch_eval_tiled<float>(srmap_problem*, const Geometry&, int, unsigned, const float*, float*,
                                      double*, int*, hipStream_t);
template int launch_eval_tiled<double>(srmap_problem*, const Geometry&, int, unsigned, const double*,
                                       double*, double*, int*, hipStream_t);

}  // namespace srmap
