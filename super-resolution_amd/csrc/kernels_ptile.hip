// kernels_ptile.hip -- the hot path as PERSISTENT TILES: one MAP gradient iteration
// (ObjectiveFunction::ComputeAllTerms, objective_function.cpp:5-20) in one launch of as many workgroups as the chip
// holds, each pulling tiles from a queue and keeping the NEXT tile's inputs in flight while it computes the current one.
//
// Same formulation, same tile and same device functions as kernels_ztile.hip (owner computes on the HR grid,
// data term objective_data_term.cpp:15-116 over image_model.cpp:86-101; TV
// tv_regularizer.cpp:110-227; BTV btv_regularizer.cpp:19-170); what changes is the life of a workgroup.  In k_eval_z
// a workgroup is born, forms and issues its requests, waits for them (40 % of its life, profiles/r02_phase_clock.txt),
// computes one tile and dies; its registers cap at 128 (two workgroups per CU must fit for the one to cover the
// other's load phase).  Here ONE workgroup per CU lives for the whole launch (two waves per SIMD, 256 registers):
//   * the x tile is double-buffered in LDS; every input register set is re-requested for the next tile right after
//     its last use for the current one (x rows after they went to LDS, observations after the residual rows, IRLS
//     weights after regulariser pass 1, the search direction after g.d), so no request is waited for inside a tile;
//   * items (border blocks first, then tiles, masked edge columns before interior ones) come from eight queues, one
//     per XCD: the tiles an XCD's CUs run at the same time are vertical neighbours and share their x halo rows in that
//     XCD's L2 (as k_eval_z's launch order arranges); a workgroup whose queue is empty steals from the next one.  The
//     queue head is a device-scope counter; the pull for the tile after next is issued one tile ahead and its result
//     is looked at one tile later (no wait);
//   * cost partials stay per TILE and are published in tile order of the index, not of the pull: the reduction is the
//     fixed-order one of k_eval_z (deterministic); a tile's partial leaves one iteration later, behind the barrier the
//     next tile needs anyway (two barriers per tile, as before).
// No MFMA: stencil path.
#include "ztile_dev.hpp"

namespace srmap {

namespace {

template <typename T, int B, int NP>
struct PArgs : ZArgs<T, B, NP> {
  unsigned* queue;  // [0..7] heads of the XCD lists, [8] exit ticket; zero between launches (the last workgroup to leave re-zeroes them)
  int nty, ntx;     // tile rows / tile columns of a channel
  int nbb;          // border blocks per channel (0: none)
  int nch;          // channels
  unsigned long long* dbg;  // development builds: per-wave phase stamps of one iteration (nullptr otherwise)
};

// Item i of list L.  List L holds the border blocks L, L + 8, ... (of all channels) and then, channel by channel and
// tile column by tile column (first, last, second, ...: the masked edge columns early), the tile rows of row band L.
// it[0]: 0 = past the end, 1 = tile (it[1] channel, it[2] tile row, it[3] tile column), 2 = border block (it[1] channel, it[2] index)
template <typename ArgsT>
__device__ __forceinline__ void p_decode(const ArgsT& A, int L, unsigned i, int (&it)[4]) {
  const int q = A.nty >> 3, rem = A.nty & 7;
  const int hL = q + (L < rem ? 1 : 0), row0 = L * q + (L < rem ? L : rem);
  const int nbt = A.nbb * A.nch;
  const int nbL = nbt > L ? (nbt - L + 7) >> 3 : 0;
  const int per_ch = A.ntx * hL;
  const unsigned nL = (unsigned)(nbL + A.nch * per_ch);
  it[0] = 0; it[1] = 0; it[2] = 0; it[3] = 0;
  if (i >= nL) return;
  if (i < (unsigned)nbL) {
    const int id = L + 8 * (int)i;
    const int ch = id / A.nbb;
    it[0] = 2; it[1] = ch; it[2] = id - ch * A.nbb;
    return;
  }
  int j = (int)i - nbL;
  const int ch = j / per_ch;
  j -= ch * per_ch;
  const int co = j / hL, r = j - co * hL;
  it[0] = 1; it[1] = ch; it[2] = row0 + r;
  it[3] = co == 0 ? 0 : (co == 1 ? A.ntx - 1 : co - 1);
}

// `raw` was pulled from list X; when that list is exhausted, pull from the others
template <typename ArgsT>
__device__ __forceinline__ void p_resolve(const ArgsT& A, int X, unsigned raw, int (&it)[4]) {
  p_decode(A, X, raw, it);
  for (int t = 1; t < 8 && it[0] == 0; ++t) {
    const int L = (X + t) & 7;
    const unsigned i = atomicAdd(&A.queue[L], 1u);
    p_decode(A, L, i, it);
  }
}

#ifdef SRMAP_DEV_INSTANCES
#define P_STAMP(k) do { if (A.dbg != nullptr && itn == 4) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) A.dbg[((size_t)blockIdx.x * NW + wv) * 16 + (k)] = t_; } } while (0)
#else
#define P_STAMP(k) do { } while (0)
#endif

template <typename T, int S, int B, int REGK, int R, bool WD, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void k_eval_p(PArgs<T, B, ZCfg<T, S, B, REGK, R, NW>::NP> A_) {
  using C = ZCfg<T, S, B, REGK, R, NW>;
  // The argument block is read through the kernel-argument segment pointer (constant address space: scalar loads) and
  // that pointer is made opaque at every phase boundary of the tile loop (P_FRESH): otherwise every argument the loop
  // touches is hoisted out of it and held in SGPRs for the whole launch -- some 200 more than exist, i.e. a
  // v_readlane / v_writelane in front of every use (190 spilled SGPRs, a quarter of a tile's time in the request burst).
  typedef const PArgs<T, B, C::NP> __attribute__((address_space(4))) KArgs;
  KArgs* Ak = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
#define A (*Ak)
#define P_FRESH() asm volatile("" : "+s"(Ak))
  constexpr int HB = C::HB, NV = C::NV, RU = C::RU;
  constexpr int ARI = (C::XR + C::NW - 1) / C::NW;  // x rows per wave
  constexpr int EXTRA = C::XC - C::CW;              // halo cells, staged by the first lanes
  constexpr int GT = 64 * (NW - 1);                 // the thread that pulls items: lane 0 of the last wave (no halo task)
  constexpr int kBorderLds = (int)((16 * sizeof(int2) + kBorderTabEntries * sizeof(ZEntry) + 16 * sizeof(double) + 7) / 8);
  __shared__ T xs2[2][C::XS_ELEMS];
  __shared__ T zs[C::ZS_ELEMS > 0 ? C::ZS_ELEMS : 1];
  __shared__ T cs[C::CS_ELEMS > 0 ? C::CS_ELEMS : 1];
  __shared__ double red[2][2][C::NW];  // [iteration parity][cost, g.d][wave]
  __shared__ T wcs[32];
  __shared__ T whs[(C::RU > 0 ? C::RU : 1) * S * C::CW];
  __shared__ int s_item[2][4];
  __shared__ double bscr[kBorderLds];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t N = (size_t)A.W * A.H;
  const size_t nl = (size_t)A.wl * A.hl;
  const bool reg_any = REGK != 0 && (A.terms & SRMAP_TERM_REG) != 0;
  const int X = (int)(__builtin_amdgcn_s_getreg(63508) & 7u);  // HW_REG_XCC_ID: the XCD this workgroup runs on (speed only)

  // ---- input registers of a tile.  ONE wait point per tile: at the top of an iteration everything requested one tile
  // ago is waited for (the compiler's s_waitcnt vmcnt(0): across a loop with conditional requests it does not count),
  // the x rows go to LDS, the other sets are copied to the registers the tile computes from, and right behind the
  // barrier that follows ALL requests of the next tile leave in one burst -- together with the previous tile's g store
  // (held back one iteration: stores count in vmcnt on gfx950 and a young one would be waited for) and the queue pull.
  // Whatever is outstanding at the wait point is therefore a whole tile old.  Every request is unconditional and goes
  // to a valid address (what a tile does not need is requested from the head of the x plane and replaced on use). ----
  T va[ARI][S], vb[ARI][S];
  T ypre[NV], ypre2[NV], wreg[S], whalo[S], dreg[S];
  T wcolv;

  const int hw2 = wv & 1;                              // waves >= 2 request (and discard) the zh halo row of wave wv & 1
  const int hrowz = hw2 == 0 ? -HB : C::TH - 1 + HB;   // halo rows of zh: tile rows -1 (wave 0) and TH (wave 1)
  const bool has_z_halo = B > 1 && A.g != nullptr && wv < 2;
  const int hrow = -(wv - 1);                          // halo rows of 2*lambda*w*r: wave 2 -> -1, wave 3 -> -2

  auto tile_edge = [&](int R0, int CJ0) -> bool {
    const int rm = A.E + HB + 1, cm = (A.E + HB + S) / S + 1;
    return (R0 - rm < 0) || (R0 + C::TH + rm > A.H) || (CJ0 - cm < 0) || (CJ0 + C::CW + cm > A.wl) || A.cr0 > 0 || A.cr1 < A.H;
  };
  auto tile_reg = [&](int R0) -> bool { return reg_any && R0 >= A.rr0 && R0 < A.rr1; };
  // which of a tile's optional inputs exist (uniform per wave except the lane tests)
  auto halo_w_ok = [&](int R0, int CJ0) -> bool {
    return tile_reg(R0) && A.w != nullptr && A.g != nullptr && RU > 0 && wv >= 2 && wv < 2 + RU && R0 + hrow >= 0 && CJ0 * S + S * lane < A.W;
  };
  auto col_w_ok = [&](int R0, int CJ0) -> bool {
    const int hgr = R0 + lane - RU, hgc = CJ0 * S - (wv == 4 ? 1 : 2);
    return tile_reg(R0) && A.w != nullptr && A.g != nullptr && RU > 0 && (wv == 4 || wv == 5) && lane < C::TH + RU && hgr >= 0 &&
           hgr < A.H && hgc >= 0;
  };
  auto own_w_ok = [&](int R0, int CJ0) -> bool {
    return tile_reg(R0) && A.w != nullptr && R0 + wv < A.H && CJ0 * S + S * lane < A.W;
  };
  auto own_d_ok = [&](int R0, int CJ0) -> bool {
    const int gr = R0 + wv;
    return gr < A.H && CJ0 * S + S * lane < A.W && gr >= A.cr0 && gr < A.cr1;
  };

  // x rows of the tile (+ halo): two row groups per wave
  auto issue_x = [&](int R0, int CJ0, int ch) {
    const T* xplane = A.x + (size_t)ch * N;
#pragma unroll
    for (int it = 0; it < ARI; ++it) {
      const int row = wv + it * C::NW;
      const int grr = R0 - C::HU + row;
      const bool row_in = row < C::XR && (unsigned)grr < (unsigned)A.H;  // uniform
      const int gca = CJ0 - C::XCL + lane, gcb = gca + C::CW;
      const bool ina = row_in && (unsigned)gca < (unsigned)A.wl;
      const bool inb = row_in && lane < EXTRA && (unsigned)gcb < (unsigned)A.wl;
      const T* sa = xplane + (ina ? (size_t)grr * A.W + (size_t)gca * S : (size_t)0);
      const T* sb = xplane + (inb ? (size_t)grr * A.W + (size_t)gcb * S : (size_t)0);
#pragma unroll
      for (int pc = 0; pc < S; ++pc) va[it][pc] = sa[pc];
#pragma unroll
      for (int pc = 0; pc < S; ++pc) vb[it][pc] = sb[pc];
    }
  };
  // IRLS weights of the halo row of 2*lambda*w*r this wave evaluates (waves 2 ..) and of its left-halo-column pixel (waves 4 / 5)
  auto issue_hw = [&](int R0, int CJ0, int ch) {
    const T* wsafe = A.w != nullptr ? A.w + (size_t)ch * N : A.x;
    const T* ph = halo_w_ok(R0, CJ0) ? wsafe + ((size_t)(R0 + hrow) * A.W + CJ0 * S + S * lane) : A.x;
#pragma unroll
    for (int pc = 0; pc < S; ++pc) whalo[pc] = ph[pc];
    const int hgr = R0 + lane - RU, hgc = CJ0 * S - (wv == 4 ? 1 : 2);
    const T* pcw = col_w_ok(R0, CJ0) ? wsafe + ((size_t)hgr * A.W + hgc) : A.x;
    wcolv = pcw[0];
  };
  // round-0 observations of the wave's row and of a zh halo row
  auto issue_y = [&](int R0, int CJ0, int ch) {
    const T* ybase = A.y + (size_t)ch * nl;
    const bool edge = tile_edge(R0, CJ0);
    z_row_prefetch<T, S, B, C>(A, wv, R0, CJ0, lane, edge, ybase, ypre);
    if (B > 1) z_row_prefetch<T, S, B, C>(A, hrowz, R0, CJ0, lane, edge, ybase, ypre2);
  };
  // IRLS weights of the thread's own pixels
  auto issue_w = [&](int R0, int CJ0, int ch) {
    const T* wsafe = A.w != nullptr ? A.w + (size_t)ch * N : A.x;
    const T* pw = own_w_ok(R0, CJ0) ? wsafe + ((size_t)(R0 + wv) * A.W + CJ0 * S + S * lane) : A.x;
#pragma unroll
    for (int pc = 0; pc < S; ++pc) wreg[pc] = pw[pc];
  };
  // WD: the search direction at the thread's pixels (g.d is produced with g)
  auto issue_d = [&](int R0, int CJ0, int ch) {
    if (!WD) return;
    const T* pd = own_d_ok(R0, CJ0) ? A.dvec + (size_t)ch * N + ((size_t)(R0 + wv) * A.W + CJ0 * S + S * lane) : A.x;
#pragma unroll
    for (int pc = 0; pc < S; ++pc) dreg[pc] = pd[pc];
  };

  // ---- items: s_item[parity] is how the pulling thread hands an item to the workgroup ----
  unsigned pend = 0;  // GT only: raw queue value of the item after next
  if (tid == GT) {
    const unsigned r0 = atomicAdd(&A.queue[X], 1u);
    int it4[4];
    p_resolve(A, X, r0, it4);
    s_item[0][0] = it4[0]; s_item[0][1] = it4[1]; s_item[0][2] = it4[2]; s_item[0][3] = it4[3];
    pend = atomicAdd(&A.queue[X], 1u);
  }
  __syncthreads();
  int c_type = __builtin_amdgcn_readfirstlane(s_item[0][0]);
  int c_ch = __builtin_amdgcn_readfirstlane(s_item[0][1]);
  int c_a = __builtin_amdgcn_readfirstlane(s_item[0][2]);
  int c_b = __builtin_amdgcn_readfirstlane(s_item[0][3]);

  T gprev[S];             // gradient of the previous tile at the thread's pixels, stored with the next request burst
  T* gdst = nullptr;
  bool g_held = false;    // uniform
#pragma unroll
  for (int pc = 0; pc < S; ++pc) gprev[pc] = T(0);
  int itn = 0;
  bool have_prev = false;  // a finished tile's wave partials wait in red[prev_par]
  int prev_par = 0;
  size_t prev_idx = 0;
  auto publish_prev = [&]() {  // thread 0, behind a barrier that follows the tile's end
    double c = 0.0, d = 0.0;
#pragma unroll
    for (int i = 0; i < C::NW; ++i) { c += red[prev_par][0][i]; if (WD) d += red[prev_par][1][i]; }
    put_partial<WD>(A, prev_idx, c, d);
  };

  while (c_type != 0) {
    // ---- border blocks (kernels_ztile.hip: what the frame-summed tiles cannot express at the image border); they
    // head every list, so a workgroup sees them before its first tile ----
    while (c_type == 2) {
      const int par = itn & 1;
      border_block<T, S, B, C::NT, WD>(A_, *A_.bd, c_a, c_ch, (void*)bscr, A_.nbb);  // (outside the tile loop: the plain argument block)
      if (tid == GT) {
        int it4[4];
        p_resolve(A, X, pend, it4);
        s_item[par ^ 1][0] = it4[0]; s_item[par ^ 1][1] = it4[1]; s_item[par ^ 1][2] = it4[2]; s_item[par ^ 1][3] = it4[3];
        pend = atomicAdd(&A.queue[X], 1u);
      }
      __syncthreads();
      if (tid == 0 && have_prev) publish_prev();
      have_prev = false;
      c_type = __builtin_amdgcn_readfirstlane(s_item[par ^ 1][0]);
      c_ch = __builtin_amdgcn_readfirstlane(s_item[par ^ 1][1]);
      c_a = __builtin_amdgcn_readfirstlane(s_item[par ^ 1][2]);
      c_b = __builtin_amdgcn_readfirstlane(s_item[par ^ 1][3]);
      ++itn;
    }
    if (c_type != 1) break;
    {  // every input of the first tile
      const int R0 = c_a * C::TH, CJ0 = c_b * C::CW;
      issue_x(R0, CJ0, c_ch); issue_hw(R0, CJ0, c_ch); issue_y(R0, CJ0, c_ch); issue_w(R0, CJ0, c_ch); issue_d(R0, CJ0, c_ch);
    }
    // =============================== tiles: (c_a, c_b) of channel c_ch ===============================
    while (c_type == 1) {
      const int par = itn & 1;
      const int R0 = c_a * C::TH, CJ0 = c_b * C::CW, C0 = CJ0 * S;
      const int ch = c_ch;
      T* xs = xs2[par];
      const int gr = R0 + wv;          // global HR row of this thread
      const int gc0 = C0 + S * lane;   // first global HR column of this thread
      const bool want_reg = tile_reg(R0);
      const T* ybase = A.y + (size_t)ch * nl;
      const bool edge = tile_edge(R0, CJ0);
      const bool reg_halo_on = want_reg && A.g != nullptr && RU > 0;
      const bool has_reg_halo = reg_halo_on && wv >= 2 && wv < 2 + RU;
      const bool col_task = reg_halo_on && (wv == 4 || wv == 5) && lane < C::TH + RU;

      P_FRESH();
      P_STAMP(0);
      // ---------------- x tile -> LDS, polyphase (scale 2^Q inside the image, 0 outside: one multiply) ----------------
#pragma unroll
      for (int it = 0; it < ARI; ++it) {
        const int row = wv + it * C::NW;
        if (row < C::XR) {  // uniform
          const int grr = R0 - C::HU + row;
          const bool row_in = (unsigned)grr < (unsigned)A.H;
          const int gca = CJ0 - C::XCL + lane, gcb = gca + C::CW;
          const T ma = (row_in && (unsigned)gca < (unsigned)A.wl) ? Pre<T>::up(T(1)) : T(0);
          const T mb = (row_in && (unsigned)gcb < (unsigned)A.wl) ? Pre<T>::up(T(1)) : T(0);
#pragma unroll
          for (int pc = 0; pc < S; ++pc) xs[row * C::XROW + pc * C::XC + lane] = va[it][pc] * ma;
          if (lane < EXTRA) {
#pragma unroll
            for (int pc = 0; pc < S; ++pc) xs[row * C::XROW + pc * C::XC + C::CW + lane] = vb[it][pc] * mb;
          }
        }
      }
      if (col_task) wcs[(wv - 4) * 16 + lane] = col_w_ok(R0, CJ0) ? wcolv : T(1);
      if (has_reg_halo) {
        const bool ok = halo_w_ok(R0, CJ0);
#pragma unroll
        for (int pc = 0; pc < S; ++pc) whs[((wv - 2) * S + pc) * C::CW + lane] = ok ? whalo[pc] : T(1);
      }
      // the register sets this tile computes from (the primary sets are re-requested behind the barrier)
      T ycur[NV], y2cur[NV], wown[S], dcur[S];
#pragma unroll
      for (int v = 0; v < NV; ++v) { ycur[v] = ypre[v]; y2cur[v] = ypre2[v]; }
      {
        const bool okw = own_w_ok(R0, CJ0), okd = WD && own_d_ok(R0, CJ0);
#pragma unroll
        for (int pc = 0; pc < S; ++pc) { wown[pc] = okw ? wreg[pc] : T(1); dcur[pc] = okd ? dreg[pc] : T(0); }
      }
      if (tid == GT) {  // the item after this one was pulled one tile ago
        int it4[4];
        p_resolve(A, X, pend, it4);
        s_item[par ^ 1][0] = it4[0]; s_item[par ^ 1][1] = it4[1]; s_item[par ^ 1][2] = it4[2]; s_item[par ^ 1][3] = it4[3];
      }
      P_STAMP(1);
      __syncthreads();  // ---- barrier A: the tile is in LDS; every wave has left the previous tile ----
      P_STAMP(2);
      P_FRESH();
      if (tid == 0 && have_prev) publish_prev();
      if (g_held && gdst != nullptr) {  // the previous tile's gradient
#pragma unroll
        for (int pc = 0; pc < S; ++pc) __builtin_nontemporal_store(gprev[pc], &gdst[pc]);  // written once, not re-read here
      }
      const int n_type = __builtin_amdgcn_readfirstlane(s_item[par ^ 1][0]);
      const int n_ch = __builtin_amdgcn_readfirstlane(s_item[par ^ 1][1]);
      const int n_a = __builtin_amdgcn_readfirstlane(s_item[par ^ 1][2]);
      const int n_b = __builtin_amdgcn_readfirstlane(s_item[par ^ 1][3]);
      // what is requested for "the next tile" when there is none: this tile again (unconditional requests, see above)
      const bool n_tile = n_type == 1;
      const int pR0 = n_tile ? n_a * C::TH : R0, pCJ0 = n_tile ? n_b * C::CW : CJ0, pch = n_tile ? n_ch : ch;
      issue_x(pR0, pCJ0, pch);
      issue_hw(pR0, pCJ0, pch);
      issue_y(pR0, pCJ0, pch);
      issue_w(pR0, pCJ0, pch);
      issue_d(pR0, pCJ0, pch);
      if (tid == GT) pend = atomicAdd(&A.queue[X], 1u);
      P_STAMP(3);
      P_FRESH();

      // in-image mask of this thread's pixels (partial tiles at the right / bottom edge)
      T mk[S];
#pragma unroll
      for (int pc = 0; pc < S; ++pc) mk[pc] = (gr < A.H && gc0 + pc < A.W) ? T(1) : T(0);
      T acc[S], zown[S];
#pragma unroll
      for (int j = 0; j < S; ++j) { acc[j] = T(0); zown[j] = T(0); }
      double cost_data = 0.0, cost_reg = 0.0;

      // ---------------- phase 1: data term ----------------
      {
        T dummy[S];
        double dcost = 0.0;
        if (edge) {
          z_row<T, S, B, C, true>(A, xs, zs, wv, R0, CJ0, lane, ybase, true, ycur, true, mk, zown, cost_data);
          if (has_z_halo) z_row<T, S, B, C, true>(A, xs, zs, hrowz, R0, CJ0, lane, ybase, true, y2cur, false, mk, dummy, dcost);
        } else {
          z_row<T, S, B, C, false>(A, xs, zs, wv, R0, CJ0, lane, ybase, true, ycur, true, mk, zown, cost_data);
          if (has_z_halo) z_row<T, S, B, C, false>(A, xs, zs, hrowz, R0, CJ0, lane, ybase, true, y2cur, false, mk, dummy, dcost);
        }
      }
      P_STAMP(4);
      P_FRESH();
      // ---------------- phase 1: regulariser ----------------
      if (want_reg) {
        const bool reg_border = (R0 + C::TH + C::WIN > A.H) || (C0 + C::TW + C::WIN > A.W);
        const bool cost_row = gr >= A.cr0 && gr < A.cr1;
        T pwl[C::NP];  // a copy: the callees take the table by (generic) reference
#pragma unroll
        for (int i = 0; i < C::NP; ++i) pwl[i] = A.powtab[i];
        if (reg_border)
          reg_row<T, S, REGK, R, C, true, true>(acc, cost_reg, xs, cs, wown, wv, lane, gr, gc0, A.W, A.H, A.lambda, pwl, A.pwsum, cost_row);
        else
          reg_row<T, S, REGK, R, C, false, true>(acc, cost_reg, xs, cs, wown, wv, lane, gr, gc0, A.W, A.H, A.lambda, pwl, A.pwsum, cost_row);
        if (has_reg_halo) {
          T dacc[S];
          double dc = 0.0;
          T whl[S];
#pragma unroll
          for (int pc = 0; pc < S; ++pc) whl[pc] = whs[((wv - 2) * S + pc) * C::CW + lane];
          if (C0 + C::TW + C::WIN > A.W || R0 + C::WIN > A.H)
            reg_row<T, S, REGK, R, C, true, false>(dacc, dc, xs, cs, whl, hrow, lane, R0 + hrow, gc0, A.W, A.H, A.lambda, pwl, A.pwsum, false);
          else
            reg_row<T, S, REGK, R, C, false, false>(dacc, dc, xs, cs, whl, hrow, lane, R0 + hrow, gc0, A.W, A.H, A.lambda, pwl, A.pwsum, false);
        }
        if (col_task) {
          const T wcol = wcs[(wv - 4) * 16 + lane];
          const int rowrel = lane - RU;
          const int lo = rowrel * C::XROW, lc = rowrel * C::CROW;  // per-lane row offsets
          if (reg_border) {
            if (RU >= 1 && wv == 4) reg_halo_col<T, S, REGK, R, C, -1, true>(xs + lo, cs + lc, wcol, rowrel, R0, C0, A.W, A.H, A.lambda, pwl);
            if (RU >= 2 && wv == 5) reg_halo_col<T, S, REGK, R, C, -2, true>(xs + lo, cs + lc, wcol, rowrel, R0, C0, A.W, A.H, A.lambda, pwl);
          } else {
            if (RU >= 1 && wv == 4) reg_halo_col<T, S, REGK, R, C, -1, false>(xs + lo, cs + lc, wcol, rowrel, R0, C0, A.W, A.H, A.lambda, pwl);
            if (RU >= 2 && wv == 5) reg_halo_col<T, S, REGK, R, C, -2, false>(xs + lo, cs + lc, wcol, rowrel, R0, C0, A.W, A.H, A.lambda, pwl);
          }
        }
      }
      P_STAMP(5);
      __syncthreads();  // ---- barrier B: zh and 2*lambda*w*r of the tile are in LDS ----
      P_STAMP(6);
      P_FRESH();

      // ---------------- phase 2 ----------------
      if (A.g != nullptr) {
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
          acc[pc] += sc * zz;
        }
      }
      if (want_reg && A.g != nullptr) {
        T pwl[C::NP];
#pragma unroll
        for (int i = 0; i < C::NP; ++i) pwl[i] = A.powtab[i];
        reg_pass2z<T, S, REGK, R, C>(acc, xs, cs, wv, lane, pwl);
      }
      P_STAMP(7);
      P_FRESH();
      // the g store is held back to the next request burst
      g_held = A.g != nullptr;
      if (A.g != nullptr) {
        const bool ok = gr < A.H && gc0 < A.W;
        gdst = ok ? A.g + (size_t)ch * N + (size_t)gr * A.W + gc0 : nullptr;  // nullptr: the thread is outside the image
#pragma unroll
        for (int pc = 0; pc < S; ++pc) gprev[pc] = acc[pc];
      }
      // ---------------- wave partials of this tile (published behind the next barrier A) ----------------
      {
        double gd = 0.0;
        if (WD) {
#pragma unroll
          for (int pc = 0; pc < S; ++pc) gd += (double)acc[pc] * (double)dcur[pc];
          gd = wave_sum_d(gd);
        }
        const double cw = wave_sum_d((double)(S * S) * cost_data + cost_reg);
        if (lane == 0) { red[par][0][wv] = cw; if (WD) red[par][1][wv] = gd; }
      }
      P_STAMP(8);
      have_prev = true;
      prev_par = par;
      prev_idx = ((size_t)ch * A.nty + c_a) * A.ntx + c_b;
      c_type = n_type; c_ch = n_ch; c_a = n_a; c_b = n_b;
      ++itn;
    }
    if (g_held && gdst != nullptr) {  // the last tile's gradient
#pragma unroll
      for (int pc = 0; pc < S; ++pc) __builtin_nontemporal_store(gprev[pc], &gdst[pc]);
    }
    g_held = false;
  }
  __syncthreads();
  if (tid == 0 && have_prev) publish_prev();
  // in-kernel finish: the last workgroup of the grid gathers the granules of the evaluation
  if (A.mfinish && blockIdx.x == gridDim.x - 1) {
    __syncthreads();
    finish_block<WD, C::NT>(A_, &red[0][0][0]);
  }
  // exit ticket: the last workgroup to leave re-arms the queues for the next launch
  if (tid == 0) {
    const unsigned old = atomicAdd(&A.queue[8], 1u);
    if (old == gridDim.x - 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) st_agent(&A.queue[i], 0u);
    }
  }
#undef A
#undef P_FRESH
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// host side
constexpr int kPersistNW = 8;
#ifdef SRMAP_DEV_INSTANCES
static unsigned long long* g_persist_dbg = nullptr;
extern "C" void srmap_dev_set_persist_dbg(void* p) { g_persist_dbg = (unsigned long long*)p; }
#endif

template <typename T, int S, int B, int REGK, int R>
static int launch_p(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g, const T* wts,
                    const ZPlan& z, double* partials, int* nblocks, bool finish_ok, bool* finished, hipStream_t st,
                    const T* dvec, double* partials_gd, bool publish) {
  using C = ZCfg<T, S, B, REGK, R, kPersistNW>;
  PArgs<T, B, C::NP> A;
  fill_common_args<T, S, B, REGK, R>(A, p, geo, obs_c0, terms, x, g, wts, z, partials, dvec, partials_gd);
  A.ntx = (geo.w + C::CW - 1) / C::CW;
  A.nty = (geo.H + C::TH - 1) / C::TH;
  A.nch = geo.C;
  A.nbb = ((terms & SRMAP_TERM_DATA) && z.n_ring > 0) ? (z.n_ring + C::NT - 1) / C::NT : 0;
  A.n_tile_partials = A.ntx * A.nty * A.nch;
  A.n_partials = A.n_tile_partials + A.nbb * A.nch;
  A.queue = z.d_queue;
  A.dbg = nullptr;
#ifdef SRMAP_DEV_INSTANCES
  A.dbg = g_persist_dbg;
#endif
  const bool finish = finish_ok && z.d_mpart != nullptr && (size_t)A.n_partials <= z.mpart_cap;
  A.mfinish = finish ? 1 : 0;  // otherwise plain partials (cost, g.d), reduced by k_finish_eval / the caller
  A.pub = (finish && publish && dvec != nullptr) ? p->eval_pub : nullptr;
  *finished = finish;
  const void* kfn = dvec != nullptr ? reinterpret_cast<const void*>(&k_eval_p<T, S, B, REGK, R, true, kPersistNW>)
                                    : reinterpret_cast<const void*>(&k_eval_p<T, S, B, REGK, R, false, kPersistNW>);
  static int occ_cache[2] = {0, 0};  // per kernel instance (this function is one template instance)
  int& occ = occ_cache[dvec != nullptr ? 1 : 0];
  if (occ == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, C::NT, 0) != hipSuccess || occ < 1)) occ = 1;
  const int cus = p->ctx->num_cus > 0 ? p->ctx->num_cus : 256;
  const long long items = (long long)A.n_partials;
  const unsigned nwg = (unsigned)std::max<long long>(1, std::min<long long>((long long)cus * occ, items));
  if (dvec != nullptr) hipLaunchKernelGGL((k_eval_p<T, S, B, REGK, R, true, kPersistNW>), dim3(nwg), dim3(C::NT), 0, st, A);
  else hipLaunchKernelGGL((k_eval_p<T, S, B, REGK, R, false, kPersistNW>), dim3(nwg), dim3(C::NT), 0, st, A);
  *nblocks = A.n_partials;
  SRMAP_HIP(p->ctx, hipGetLastError());
  return SRMAP_OK;
}

// (scale, blur size, fused regulariser kind, BTV range) instances
#ifdef SRMAP_DEV_INSTANCES
#define SRMAP_PERSIST_INSTANCES(X) X(4, 3, 2, 3)
#else
#define SRMAP_PERSIST_INSTANCES(X) X(4, 3, 2, 3) X(4, 1, 2, 3) X(3, 1, 1, 0) X(2, 1, 1, 0)
#endif

bool persist_has_instance(int S, int B, int regk, int regr) {
#define X(s_, b_, k_, r_) if (S == s_ && B == b_ && (regk == 0 || (regk == k_ && regr == r_))) return true;
  SRMAP_PERSIST_INSTANCES(X)
#undef X
  return false;
}

template <typename T>
int launch_eval_persist(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms, const T* x, T* g, const T* wts,
                        int regk, int regr, double* partials, int* nblocks, bool finish, bool* fin, hipStream_t st,
                        const T* dv, double* pgd, bool publish) {
  const ZPlan& z = *static_cast<const ZPlan*>(p->zplan);
  const int S = geo.s, B = geo.b;
  // regk == 0 (the fused regulariser is not part of this evaluation): any instance of the geometry serves, its
  // regulariser switched off by `terms`
#define X(s_, b_, k_, r_)                                                                                              \
  if (S == s_ && B == b_ && (regk == 0 || (regk == k_ && regr == r_)))                                                  \
    return launch_p<T, s_, b_, k_, r_>(p, geo, obs_c0, terms, x, g, wts, z, partials, nblocks, finish, fin, st, dv, pgd, publish);
  SRMAP_PERSIST_INSTANCES(X)
#undef X
  return set_error(p->ctx, SRMAP_EUNSUPPORTED, "no persistent tile kernel for scale %d blur %d regulariser %d/%d", S, B, regk, regr);
}

template int launch_eval_persist<float>(srmap_problem*, const Geometry&, int, unsigned, const float*, float*, const float*,
                                        int, int, double*, int*, bool, bool*, hipStream_t, const float*, double*, bool);
template int launch_eval_persist<double>(srmap_problem*, const Geometry&, int, unsigned, const double*, double*,
                                         const double*, int, int, double*, int*, bool, bool*, hipStream_t, const double*,
                                         double*, bool);

template <typename T, int S, int B, int REGK, int R>
static void preload_p() {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_eval_p<T, S, B, REGK, R, false, kPersistNW>));
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_eval_p<T, S, B, REGK, R, true, kPersistNW>));
}
void persist_preload(const srmap_problem* p) {
  const ZPlan* z = static_cast<const ZPlan*>(p->zplan);
  if (!z || z->subpix) return;
#define X(s_, b_, k_, r_)                                                             \
  if (z->S == s_ && z->B == b_ && z->regk == k_ && (k_ != 2 || z->regr == r_)) {       \
    if (p->dtype == SRMAP_F32) preload_p<float, s_, b_, k_, r_>();                     \
    else preload_p<double, s_, b_, k_, r_>();                                          \
  }
  SRMAP_PERSIST_INSTANCES(X)
#undef X
}

}  // namespace srmap
