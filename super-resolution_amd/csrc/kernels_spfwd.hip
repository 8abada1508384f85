// kernels_spfwd.hip -- forward model A_k x - y_k for SUB-PIXEL shifts as a tile kernel.
//
// Reference path: ImageModel::ApplyToImage (image_model.cpp:86-91) = MotionModule (warpAffine, bilinear with the
// 1/32-px table, zero border; motion_module.cpp:18-38) -> BlurModule (filter2D, BORDER_CONSTANT;
// blur_module.cpp:24-28) -> DownsamplingModule (nearest, downsampling_module.cpp:20-33), minus the observation
// (objective_data_term.cpp:29-50).  It feeds the sub-pixel form of the tile kernel (kernels_ztile.hip, SP = true):
// the residuals land in one [K][C][h][w] buffer that k_eval_z gathers through its 4-tap phase tables.
//
// One workgroup = kLRH LR rows x 64 LR cells of one channel, ALL frames.  The HR window those residuals read
// (S*kLRH rows plus the span of the frames' integer offsets plus blur + bilinear reach) is staged once in LDS in the
// polyphase layout xs[row][col mod S][cell] (conflict-free for lanes = consecutive LR cells), zero outside the
// image -- which IS warpAffine's zero border.  Blur and bilinear taps of a frame collapse into one
// (b+1) x (b+1) stencil (built on the host per frame, read through scalar loads): 16 LDS reads and 16 FMAs per
// residual for the 3 x 3 blur.  filter2D's constant border touches only LR row 0 / column 0 (blur taps on HR
// row / column -1 are dropped even when their warp source is inside the image): those threads evaluate the
// unfused blur-of-bilinear form with the dropped taps masked.
// HBM-bound: reads x once (+ window halo from L2), y once, writes the residuals once.
#include <algorithm>
#include <climits>
#include <vector>

#include "cg_norm.hpp"
#include "srmap_internal.hpp"

namespace srmap {

namespace {

constexpr int kLRH = 4;   // LR rows per workgroup
constexpr int kCW = 64;   // LR cells per row = lanes
constexpr int kNW = 8;    // waves: LR row = wave % kLRH, frame group = wave / kLRH
constexpr int kNFG = kNW / kLRH;
constexpr int kMaxRowsPerWave = 4;

template <typename T>
struct SpFrame {  // one frame's forward warp folded with the blur
  int oy, ox;     // integer part of the bilinear gather (source = pixel + (oy, ox) + {0,1}^2)
  int rowb;       // window row of stencil row 0 for LR row 0 of the workgroup: -HB + oy - RLO
  int pad0;
  int coloff[4];  // per stencil column e1: window offset (column phase) * XC + (cell offset - CLO), worked out on the
                  // host (the divisions and the 64-bit address arithmetic per frame and column ran on the scalar unit)
  T comb[4][16];  // (b+1) x (b+1) stencils, row major with stride b+1: [0] interior, [1] without blur column 0
                  // (LR column 0), [2] without blur row 0 (LR row 0), [3] without both
};

template <typename T>
struct SpfArgs {
  const T* x;
  const T* y;
  T* out;
  double* partials;
  const SpFrame<T>* frames;
  int K, W, H, wl, hl, C, obs_C, obs_c0, cr0, cr1;
  int RLO, CLO, XR, XC;  // window: rows S*i0 + RLO .. + XR-1, cells j0 + CLO .. + XC-1
  double cost_scale;
  // FOLD instances (solver line search): the point is fold_xk + fold_stp * d, formed as the window is loaded -- the
  // expression of solver.hip's k_axpy_out, the same contraction -- and the workgroup's OWN pixels (the S kLRH x S kCW
  // block of its LR cells) are written to `xout`: the tile kernel behind this launch, and the solver, find the trial
  // point there.  The window holds what the frames read, which need not include every own pixel (frames that all read
  // up-left of their LR pixel leave the block's last row out): FOLD instances walk the union of window and own block,
  // rows S*i0 + RF0 .. + NRF-1, cells j0 + CF0 .. + NCF-1, and stage only the window's part in LDS.
  // dvec / fold_norms as in ZArgs (cg_norm.hpp).
  int RF0, NRF, CF0, NCF;
  const T* fold_xk;
  const T* dvec;
  T* xout;
  T fold_stp;
  const double* fold_norms;
};

__device__ __forceinline__ int fdiv_rt(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

template <typename T, int S, int B, bool FOLD>
__global__ __launch_bounds__(64 * kNW) void k_forward_sp(SpfArgs<T> A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* xs = reinterpret_cast<T*>(smem_raw);
  __shared__ double red[kNW];
  constexpr int NB1 = B + 1, HB = (B - 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = wv % kLRH, par = wv / kLRH;
  const int i0 = blockIdx.y * kLRH, j0 = blockIdx.x * kCW, ch = blockIdx.z;
  const size_t N = (size_t)A.W * A.H, nl = (size_t)A.wl * A.hl;
  const T* xplane = (FOLD ? A.fold_xk : A.x) + (size_t)ch * N;
  const int XROW = S * A.XC;
  const DirScale dsc = dir_scale(FOLD ? A.fold_norms : nullptr);

  // ---- window -> LDS (all loads first) ----
  T va[kMaxRowsPerWave][S], vb[kMaxRowsPerWave][S];
  // the rows / cells this workgroup walks: the window; FOLD: the union of window and own block (see SpfArgs)
  const int r_lo = FOLD ? A.RF0 : A.RLO, n_rows = FOLD ? A.NRF : A.XR, c_lo = FOLD ? A.CF0 : A.CLO, n_cells = FOLD ? A.NCF : A.XC;
  const bool has_b = lane + kCW < n_cells;
#pragma unroll
  for (int it = 0; it < kMaxRowsPerWave; ++it) {
    const int row = wv + it * kNW;
    const int grr = S * i0 + r_lo + row;
    const bool row_in = row < n_rows && (unsigned)grr < (unsigned)A.H;  // uniform
    const int gca = j0 + c_lo + lane, gcb = gca + kCW;
    const bool ina = row_in && (unsigned)gca < (unsigned)A.wl;
    const bool inb = row_in && has_b && (unsigned)gcb < (unsigned)A.wl;
    const T* sa = xplane + (ina ? (size_t)grr * A.W + (size_t)gca * S : (size_t)0);
    const T* sb = xplane + (inb ? (size_t)grr * A.W + (size_t)gcb * S : (size_t)0);
#pragma unroll
    for (int pc = 0; pc < S; ++pc) va[it][pc] = sa[pc];
#pragma unroll
    for (int pc = 0; pc < S; ++pc) vb[it][pc] = sb[pc];
    if (FOLD) {
      const T* dpl = A.dvec + (size_t)ch * N;
      const T* da = dpl + (sa - xplane);
      const T* db = dpl + (sb - xplane);
      T vda[S], vdb[S];
#pragma unroll
      for (int pc = 0; pc < S; ++pc) vda[pc] = da[pc];
#pragma unroll
      for (int pc = 0; pc < S; ++pc) vdb[pc] = db[pc];
#pragma unroll
      for (int pc = 0; pc < S; ++pc) {
        va[it][pc] = va[it][pc] + A.fold_stp * dir_elem<T>(vda[pc], dsc);
        vb[it][pc] = vb[it][pc] + A.fold_stp * dir_elem<T>(vdb[pc], dsc);
      }
      // the workgroup's own pixels of the trial point go out (every image pixel is some workgroup's own exactly once)
      const int orow = r_lo + row;  // HR row relative to S * i0
      const bool own_row = row_in && orow >= 0 && orow < S * kLRH;
      T* xo = A.xout + (size_t)ch * N;
      if (own_row && ina && gca >= j0 && gca < j0 + kCW) {
#pragma unroll
        for (int pc = 0; pc < S; ++pc) xo[(size_t)grr * A.W + (size_t)gca * S + pc] = va[it][pc];
      }
      if (own_row && inb && gcb >= j0 && gcb < j0 + kCW) {
#pragma unroll
        for (int pc = 0; pc < S; ++pc) xo[(size_t)grr * A.W + (size_t)gcb * S + pc] = vb[it][pc];
      }
    }
#pragma unroll
    for (int pc = 0; pc < S; ++pc) {
      va[it][pc] = ina ? va[it][pc] : T(0);
      vb[it][pc] = inb ? vb[it][pc] : T(0);
    }
  }
#pragma unroll
  for (int it = 0; it < kMaxRowsPerWave; ++it) {
    const int row = wv + it * kNW;
    if (!FOLD) {
      if (row < A.XR) {  // uniform
#pragma unroll
        for (int pc = 0; pc < S; ++pc) xs[row * XROW + pc * A.XC + lane] = va[it][pc];
        if (has_b) {
#pragma unroll
          for (int pc = 0; pc < S; ++pc) xs[row * XROW + pc * A.XC + kCW + lane] = vb[it][pc];
        }
      }
    } else {
      const int wrow = r_lo + row - A.RLO;                   // window row (uniform)
      const int wca = c_lo + lane - A.CLO, wcb = wca + kCW;  // window cells
      if (row < n_rows && wrow >= 0 && wrow < A.XR) {
        if (wca >= 0 && wca < A.XC) {
#pragma unroll
          for (int pc = 0; pc < S; ++pc) xs[wrow * XROW + pc * A.XC + wca] = va[it][pc];
        }
        if (has_b && wcb >= 0 && wcb < A.XC) {
#pragma unroll
          for (int pc = 0; pc < S; ++pc) xs[wrow * XROW + pc * A.XC + wcb] = vb[it][pc];
        }
      }
    }
  }
  __syncthreads();

  const int i = i0 + li, j = j0 + lane;
  const bool valid = i < A.hl && j < A.wl;
  const bool cost_row = (i * S >= A.cr0 && i * S < A.cr1);
  // filter2D's constant border: only LR row 0 / column 0 own blur taps outside the image (HB < S); they use the
  // stencils built without those taps
  const int vrow = (HB > 0 && i == 0) ? 2 : 0;  // uniform per wave
  const bool col0_block = HB > 0 && j0 == 0;    // uniform
  double sq = 0.0;
  const size_t lp = (size_t)i * A.wl + j;
  constexpr int U = 4;  // frames per round: their observations are requested together
  for (int kb = par; kb < A.K; kb += kNFG * U) {
    T yv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = kb + kNFG * u;
      yv[u] = (valid && k < A.K) ? A.y[((size_t)k * A.obs_C + ch + A.obs_c0) * nl + lp] : T(0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = kb + kNFG * u;
      if (k >= A.K) break;  // uniform
      // constant address space: uniform -> scalar loads even after the residual stores of earlier frames
      typedef const SpFrame<T> __attribute__((address_space(4))) * FramePtr;
      FramePtr fp = (FramePtr)(unsigned long long)(A.frames + k);
      const auto& f = *fp;
      const int roff = S * li + f.rowb;  // window row of stencil row 0
      T acc = T(0);
      const auto* cm = f.comb[vrow];
      const auto* cmc = f.comb[vrow | 1];
#pragma unroll
      for (int e1 = 0; e1 < NB1; ++e1) {
        const T* col = xs + roff * XROW + f.coloff[e1] + lane;
#pragma unroll
        for (int a1 = 0; a1 < NB1; ++a1) acc += cm[a1 * NB1 + e1] * col[a1 * XROW];
      }
      if (col0_block && lane == 0) {  // LR column 0: again with the stencil that drops blur column 0
        acc = T(0);
#pragma unroll
        for (int e1 = 0; e1 < NB1; ++e1) {
          const T* col = xs + roff * XROW + f.coloff[e1];
#pragma unroll
          for (int a1 = 0; a1 < NB1; ++a1) acc += cmc[a1 * NB1 + e1] * col[a1 * XROW];
        }
      }
      const T res = acc - yv[u];
      if (valid) {
        A.out[((size_t)k * A.C + ch) * nl + lp] = res;
        if (cost_row) sq += (double)res * (double)res;
      }
    }
  }
  sq = wave_sum64(sq);
  if (lane == 0) red[wv] = sq;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kNW; ++w) s += red[w];
    A.partials[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = A.cost_scale * s;
  }
}

template <typename T>
bool upload_frames(const srmap_problem* p, SpForwardPlan* sp) {
  const Geometry& g = p->geo;
  const int B = g.b, NB1 = B + 1;
  std::vector<SpFrame<T>> fr((size_t)g.K);
  for (int k = 0; k < g.K; ++k) {
    const WarpTaps<double>& f = p->fwd_warps[k];
    SpFrame<T>& o = fr[k];
    o.oy = f.oy; o.ox = f.ox; o.pad0 = 0;
    o.rowb = -g.hb + f.oy - sp->RLO;
    for (int e1 = 0; e1 < 4; ++e1) {
      const int co = f.ox - g.hb + e1;
      const int cq = co >= 0 ? co / g.s : -((-co + g.s - 1) / g.s), ph = co - cq * g.s;
      o.coloff[e1] = ph * sp->XC + (cq - sp->CLO);
    }
    T w[4];
    for (int t = 0; t < 4; ++t) w[t] = (T)(t < f.ntaps ? f.w[t] : 0.0);
    for (int v = 0; v < 4; ++v) {
      const int a_lo = (v & 2) && g.hb > 0 ? g.hb : 0, e_lo = (v & 1) && g.hb > 0 ? g.hb : 0;  // dropped blur rows / columns
      for (int q = 0; q < 16; ++q) o.comb[v][q] = T(0);
      for (int a1 = 0; a1 < NB1; ++a1)
        for (int e1 = 0; e1 < NB1; ++e1) {
          T s = T(0);
          for (int t = 0; t < 4; ++t) {
            const int a = a1 - (t >> 1), e = e1 - (t & 1);
            if (a >= a_lo && a < B && e >= e_lo && e < B) s += (T)p->blur2d[(size_t)a * B + e] * w[t];
          }
          o.comb[v][a1 * NB1 + e1] = s;
        }
    }
  }
  if (hipMalloc(&sp->d_frames, sizeof(SpFrame<T>) * fr.size()) != hipSuccess) return false;
  return hipMemcpy(sp->d_frames, fr.data(), sizeof(SpFrame<T>) * fr.size(), hipMemcpyHostToDevice) == hipSuccess;
}

template <typename T, int S, int B>
int launch_typed(srmap_problem* p, const Geometry& geo, const SpForwardPlan& sp, const T* x, const T* y, int obs_C,
                 int obs_c0, T* out, double* partials, int* nblocks, hipStream_t st, const SpFold& fold) {
  SpfArgs<T> A;
  A.fold_xk = (const T*)fold.xk; A.dvec = (const T*)fold.dvec; A.xout = const_cast<T*>(x); A.fold_stp = (T)fold.stp;
  A.fold_norms = fold.norms;
  if (fold.xk != nullptr && !sp.can_fold)
    return set_error(p->ctx, SRMAP_EINVAL, "internal: the forward tile kernel cannot form the trial point for this geometry (no fold)");
  A.RF0 = sp.RF0; A.NRF = sp.NRF; A.CF0 = sp.CF0; A.NCF = sp.NCF;
  A.x = x; A.y = y; A.out = out; A.partials = partials;
  A.frames = (const SpFrame<T>*)sp.d_frames;
  A.K = geo.K; A.W = geo.W; A.H = geo.H; A.wl = geo.w; A.hl = geo.h; A.C = geo.C;
  A.obs_C = obs_C; A.obs_c0 = obs_c0; A.cr0 = geo.cr0; A.cr1 = geo.cr1;
  A.RLO = sp.RLO; A.CLO = sp.CLO; A.XR = sp.XR; A.XC = sp.XC;
  A.cost_scale = (double)geo.s * (double)geo.s;
  dim3 grid((unsigned)((geo.w + kCW - 1) / kCW), (unsigned)((geo.h + kLRH - 1) / kLRH), (unsigned)geo.C);
  const size_t lds = (size_t)sp.XR * S * sp.XC * sizeof(T);
  if (fold.xk != nullptr) hipLaunchKernelGGL((k_forward_sp<T, S, B, true>), grid, dim3(64 * kNW), lds, st, A);
  else hipLaunchKernelGGL((k_forward_sp<T, S, B, false>), grid, dim3(64 * kNW), lds, st, A);
  SRMAP_HIP(p->ctx, hipGetLastError());
  *nblocks = (int)(grid.x * grid.y * grid.z);
  return SRMAP_OK;
}

template <typename T, int S, int B>
void preload_typed() {
  hipFuncAttributes attr;
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_forward_sp<T, S, B, false>));
  (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&k_forward_sp<T, S, B, true>));
}

}  // namespace

bool spfwd_plan(srmap_problem* p, SpForwardPlan* sp) {
  spfwd_release(sp);
  const Geometry& g = p->geo;
  const int S = g.s, B = g.b;
  if (!p->has_motion || !p->maps_regular || S < 2 || S > 4 || (B != 1 && B != 3)) return false;
  int omin = INT_MAX, omax = INT_MIN;
  for (int k = 0; k < g.K; ++k) {
    const WarpTaps<double>& f = p->fwd_warps[k];
    if (f.ytab != nullptr || (f.ntaps != 1 && f.ntaps != 4)) return false;
    omin = std::min(omin, std::min(f.ox, f.oy));
    omax = std::max(omax, std::max(f.ox, f.oy));
  }
  const int hb = g.hb;
  sp->RLO = omin - hb;
  sp->XR = S * (kLRH - 1) + (omax - omin) + B + 1;
  sp->CLO = (omin - hb >= 0) ? (omin - hb) / S : -((hb - omin + S - 1) / S);
  const int chi = omax - hb + B;  // rightmost column offset of the last cell's stencil
  const int cq = chi >= 0 ? chi / S : -((-chi + S - 1) / S);
  sp->XC = kCW + cq - sp->CLO;
  const size_t lds = (size_t)sp->XR * S * sp->XC * (p->dtype == SRMAP_F32 ? 4 : 8);
  if (sp->XR > kMaxRowsPerWave * kNW || sp->XC > 2 * kCW || sp->XC < kCW || lds > 64 * 1024) return false;
  const bool ok = p->dtype == SRMAP_F32 ? upload_frames<float>(p, sp) : upload_frames<double>(p, sp);
  if (!ok) { spfwd_release(sp); return false; }
  sp->ok = true;
  // fold (solver line search): a workgroup walks the union of its window and its own S kLRH x S kCW pixels
  sp->RF0 = std::min(sp->RLO, 0); sp->NRF = std::max(sp->RLO + sp->XR, S * kLRH) - sp->RF0;
  sp->CF0 = std::min(sp->CLO, 0); sp->NCF = std::max(sp->CLO + sp->XC, kCW) - sp->CF0;
  sp->can_fold = sp->NRF <= kMaxRowsPerWave * kNW && sp->NCF <= 2 * kCW;
  if (p->dtype == SRMAP_F32) {
    if (S == 2 && B == 1) preload_typed<float, 2, 1>(); else if (S == 2) preload_typed<float, 2, 3>();
    else if (S == 3 && B == 1) preload_typed<float, 3, 1>(); else if (S == 3) preload_typed<float, 3, 3>();
    else if (B == 1) preload_typed<float, 4, 1>(); else preload_typed<float, 4, 3>();
  } else {
    if (S == 2 && B == 1) preload_typed<double, 2, 1>(); else if (S == 2) preload_typed<double, 2, 3>();
    else if (S == 3 && B == 1) preload_typed<double, 3, 1>(); else if (S == 3) preload_typed<double, 3, 3>();
    else if (B == 1) preload_typed<double, 4, 1>(); else preload_typed<double, 4, 3>();
  }
  return true;
}

void spfwd_release(SpForwardPlan* sp) {
  if (sp->d_frames) (void)hipFree(sp->d_frames);
  sp->d_frames = nullptr;
  sp->ok = false;
}

template <typename T>
int launch_forward_sp(srmap_problem* p, const Geometry& geo, const SpForwardPlan& sp, const T* x, const T* y,
                      int obs_C, int obs_c0, T* out, double* partials, int* nblocks, hipStream_t st, const SpFold& fold) {
  const int S = geo.s, B = geo.b;
#define SPF(SS, BB) \
  if (S == SS && B == BB) return launch_typed<T, SS, BB>(p, geo, sp, x, y, obs_C, obs_c0, out, partials, nblocks, st, fold)
  SPF(2, 1); SPF(2, 3); SPF(3, 1); SPF(3, 3); SPF(4, 1); SPF(4, 3);
#undef SPF
  return set_error(p->ctx, SRMAP_EUNSUPPORTED, "no forward tile kernel for scale %d blur %d", S, B);
}

template int launch_forward_sp<float>(srmap_problem*, const Geometry&, const SpForwardPlan&, const float*,
                                      const float*, int, int, float*, double*, int*, hipStream_t, const SpFold&);
template int launch_forward_sp<double>(srmap_problem*, const Geometry&, const SpForwardPlan&, const double*,
                                       const double*, int, int, double*, double*, int*, hipStream_t, const SpFold&);

}  // namespace srmap
