// srmap_internal.hpp -- shared declarations of libsrmap.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "srmap.h"

namespace srmap {

constexpr int kMaxRegularizers = 4;
constexpr int kMaxBtvRange = 8;        // alpha^(i+j) table holds 2*range+1 entries
constexpr int kMaxBlurTaps = 15 * 15;  // b*b taps kept in kernel-argument space

// One MotionModule warp (forward or transpose) of one frame, as cv::warpAffine
// evaluates it (motion_module.cpp:18-51): source pixel = destination + (ox, oy)
// plus up to four bilinear taps with 1/32-pixel quantised weights.
template <typename T>
struct WarpTaps {
  int ox, oy;
  int ntaps;  // 1 = integer shift (weights 1,0,0,0), 4 = bilinear
  int fx;     // x fraction index 0..31 (1/32 px) -- used with ytab
  T w[4];     // (0,0) (1,0) (0,1) (1,1) as (dx,dy) tap offsets
  // warpAffine evaluates the y coordinate of every row in floating point before quantising it to 1/32 px; for a dy
  // within rounding distance of a quantisation tie the fraction index differs from row to row.  Then ytab (device,
  // one int per destination row: source row << 5 | fraction index) replaces oy and the y half of w.
  const int* ytab;
};

struct Geometry {
  int W, H, C, K;  // HR size, channels, frames
  int w, h;        // LR size
  int s;           // scale
  int b, hb;       // blur kernel size (1 = none) and (b-1)/2
  int rr0, rr1;    // HR rows [rr0, rr1) whose regulariser terms (gradient AND cost) this evaluation produces: frame
                   // sharding splits the regulariser over the ranks by row band (multiples of 8, the tile height;
                   // tile kernels only; default 0, H)
  int cr0, cr1;    // HR rows [cr0, cr1) whose cost terms are counted (row-band sharding; default 0, H):
                   // regulariser pixels of those rows, data residuals of LR rows [cr0/s, cr1/s)
  int zlo, zhi;    // channel sharding: a halo plane exists before channel 0 / after channel C-1 of this view
                   // (3-D TV couples across it, tv_regularizer.cpp:205-222); 0 otherwise
};

struct RegSpec {
  int kind;
  int range;
  double decay;
  double lambda;
  void* weights;  // device [C][H][W] dtype; nullptr = all ones
  double pow_table[2 * kMaxBtvRange + 1];  // std::pow(decay, k), host libm
};

}  // namespace srmap

struct srmap_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string error;
  int num_cus = 0;
  // pinned host staging: two chunks for pipelined host<->device copies of caller (pageable) buffers,
  // and a small scalar block the reduction kernels write directly (no copy kernels, one sync)
  void* h_stage[2] = {nullptr, nullptr};
  hipEvent_t h_event[2] = {nullptr, nullptr};
  double* h_scal = nullptr;  // [16], host-mapped
  void* blas = nullptr;      // rocblas_handle of this context (channel_map.hip), created on first use
};

struct srmap_problem {
  srmap_ctx* ctx = nullptr;
  srmap::Geometry geo{};
  int dtype = SRMAP_F64;
  int impl = SRMAP_IMPL_AUTO;
  bool has_motion = false;
  bool maps_regular = true;       // decimation map == s*j on both axes
  std::vector<double> shifts;     // K x 2
  std::vector<double> blur2d;     // b*b (double); transposed copy in blur2d_t
  std::vector<double> blur2d_t;
  std::vector<double> blur1d;     // b (the separable factor: blur2d = blur1d * blur1d^T)
  // device constants
  std::vector<int*> d_ytabs;      // per-row y tables of frames whose warpAffine y table is not uniform (owned)
  void* d_fwd_warps = nullptr;    // WarpTaps<T>[K]
  void* d_bwd_warps = nullptr;    // WarpTaps<T>[K]
  void* d_blur = nullptr;         // T[b*b]
  void* d_blur_t = nullptr;       // T[b*b]
  int* d_col_map = nullptr;       // int[w]  decimation source column
  int* d_row_map = nullptr;       // int[h]
  // host mirrors of the warp taps (double) for tile planning
  std::vector<srmap::WarpTaps<double>> fwd_warps, bwd_warps;
  // state
  void* d_obs = nullptr;          // [K][C][h][w] dtype
  bool have_obs = false;
  void* d_resid = nullptr;        // [K][C][h][w] dtype scratch
  void* d_regvals = nullptr;      // [C][H][W] dtype scratch
  void* d_x = nullptr;            // [C][H][W] staging for host-buffer entry points
  void* d_g = nullptr;
  void* d_tmp = nullptr;          // [C][H][W] staging (gradient constants, values)
  double* d_partials = nullptr;   // per-block cost partials
  size_t partials_cap = 0;
  double* d_cost = nullptr;       // [8] reduced scalars: [0] cost, [1] g.d when gd_valid
  const void* eval_dvec = nullptr;  // set by the solver around an evaluation: direction d (device, dtype); the tile
                                    // kernel then produces g.d with the gradient (one pass and two launches fewer)
  bool gd_valid = false;            // the last evaluation left g.d in d_cost[1]
  // set by the solver's line search around an evaluation (with eval_dvec): the point to evaluate is
  // eval_fold_xk + eval_fold_stp * eval_dvec, formed by the tile kernel as it loads its window and written to the x the
  // evaluation was given (no separate n-vector pass per trial point); only where ztile_can_fold() says so
  const void* eval_fold_xk = nullptr;
  double eval_fold_stp = 0.0;
  const double* eval_fold_norms = nullptr;  // device {max|dk|, dk.dk}: eval_dvec is the unnormalised direction (solver.hip norm_elem)
  // set by the solver around an evaluation: host-mapped words the evaluation's finish kernel publishes
  // {cost, g.d} to, followed by the arrival tag (saves the separate publish launch); eval_published reports it did
  double* eval_pub = nullptr;
  double* eval_pub_tag_slot = nullptr;
  double selfcheck_beta_den = 0.0;   // largest relative deviation of the derived beta denominator from the directly summed one
  double* eval_timeout_host = nullptr;   // host-mapped word the in-kernel finish raises when it gives up waiting (solver)
  double eval_pub_tag = 0.0;
  bool eval_published = false;
  // stream ordering of the device STATE an evaluation reads (observations, IRLS weights): state_ev is recorded on the
  // stream that last wrote it asynchronously (state_stream); an evaluation on another stream waits for it once
  // (state_seen); a writer on another stream than the last evaluation's (use_stream) drains that stream first
  hipEvent_t state_ev = nullptr;
  hipStream_t state_stream = nullptr, state_seen = nullptr, use_stream = nullptr;
  // set by the row-sharded evaluation around one evaluation: ov_hook(ov_arg) posts the halo exchange of x (on another
  // stream) and records ov_event when the halo rows [0, ov_top) and [H - ov_bot, H) are in place.  The tile kernels
  // run the tiles that read no halo row first, then the hook, then -- behind the event -- the remaining tile rows;
  // every other path calls the hook and waits before it starts.
  int (*ov_hook)(void*) = nullptr;
  void* ov_arg = nullptr;
  hipEvent_t ov_event = nullptr;
  int ov_top = 0, ov_bot = 0;
  // frame sharding: whether EVERY rank of the communicator can evaluate the regulariser of a row band (agreed once by an
  // all-reduce, solver.hip shard_eval); the key it was agreed for
  const void* band_comm = nullptr;
  unsigned long long plan_gen = 1;   // bumped whenever the tile plan or the implementation choice changes (a freed and
                                     // re-allocated plan can come back at the same address: pointer identity is no key)
  unsigned long long band_gen = 0;   // generation the agreement below was reached for
  unsigned band_terms = 0;
  bool band_all = false;
  int nreg = 0;
  srmap::RegSpec reg[srmap::kMaxRegularizers];
  void* zplan = nullptr;          // srmap::ZPlan of the z-tile kernels (kernels_ztile.hip), owned; nullptr = not covered
  // channel view of the current evaluation (split_channels solves one channel
  // at a time, irls_map_solver.cpp:200-262); default = all channels
  int view_c0 = 0, view_C = 0;
  bool view_coupled = false;      // the view's neighbour planes are halo channels of a channel shard (3-D TV reads them)
  size_t elem() const { return dtype == SRMAP_F32 ? 4 : 8; }
  size_t hr_count() const { return (size_t)geo.C * geo.H * geo.W; }
  size_t lr_count() const { return (size_t)geo.K * geo.C * geo.h * geo.w; }
};

namespace srmap {

int set_error(srmap_ctx* ctx, int status, const char* fmt, ...);
void blas_release(srmap_ctx* ctx);

#define SRMAP_HIP(ctx, call)                                                     \
  do {                                                                           \
    hipError_t e_ = (call);                                                      \
    if (e_ != hipSuccess)                                                        \
      return ::srmap::set_error((ctx), SRMAP_EHIP, "%s failed: %s (%s:%d)", #call, \
                                hipGetErrorString(e_), __FILE__, __LINE__);      \
  } while (0)

// ---- kernel launchers (kernels_direct.hip) ----
// out = A_k x (y == nullptr) or A_k x - y_k for frames [k0, k0+nk); optional
// cost partials (s^2 * sum of squares, double) appended at partials[0..nblocks).
// `g` is the geometry of this evaluation (g.C may be a channel sub-range of the
// problem: y is indexed with the problem's channel count obs_C and offset obs_c0).
template <typename T>
int launch_forward_direct(srmap_problem* p, const Geometry& g, const T* x, const T* y,
                          int obs_C, int obs_c0, T* out, int k0, int nk,
                          double* partials, int* nblocks, hipStream_t st);
// g = (accumulate ? g : 0) + 2 s^2 sum_k A_k^T r_k   (r: [K][C][h][w])
template <typename T>
int launch_gather_direct(srmap_problem* p, const Geometry& geo, const T* resid, T* g,
                         int k0, int nk, double out_scale, bool accumulate,
                         hipStream_t st, int ring = 0, T* ringbuf = nullptr);
// whether the ring mode (ring > 0) runs as k_gather_ring (which can also write the ring's values to a side buffer)
bool gather_ring_kernel_ok(const srmap_problem* p, const Geometry& geo, int nk, int ring);
template <typename T>
int launch_reg_values(srmap_problem* p, const Geometry& geo, const RegSpec& rs,
                      const T* x, T* values, hipStream_t st);
// g += d(reg)/dx with constants c = lambda_or_1 * gc[p]; optional cost partials
// lambda * w * r^2 (only meaningful when gc are the IRLS weights).
template <typename T>
int launch_reg_gradient_direct(srmap_problem* p, const Geometry& geo, const RegSpec& rs,
                               const T* x, const T* gc, double gc_scale, const T* values,
                               T* g, bool accumulate, double* partials, int* nblocks,
                               hipStream_t st);
template <typename T>
int launch_irls_weights(srmap_problem* p, const T* values, T* weights, size_t n,
                        hipStream_t st);
template <typename T>
int launch_reg_weights(srmap_problem* p, const Geometry& geo, const RegSpec& rs,
                       const T* x, T* weights, hipStream_t st);
int reduce_scratch_slots(size_t n);
int launch_reduce_partials(srmap_problem* p, const double* partials, int n,
                           double* out, hipStream_t st);

// ---- z-tile kernels (kernels_ztile.hip): the hot path ----
bool ztile_plan(srmap_problem* p);
void ztile_release(srmap_problem* p);
void ztile_preload(const srmap_problem* p);
void ztile_rearm(srmap_problem* p);  // re-initialise the granules of the in-kernel cost reduction (after its time-out)
bool ztile_overlaps_halo(const srmap_problem* p);  // the next tile evaluation can run interior tiles under the halo exchange
bool ztile_reg_band_ok(const srmap_problem* p, unsigned terms);
bool ztile_can_fold(const srmap_problem* p);  // a TERM_ALL evaluation with eval_dvec can form its point from xk + stp * d itself  // the tile kernel alone produces the regulariser part
size_t ztile_partials_needed(const srmap_problem* p);
template <typename T>
int launch_eval_ztile(srmap_problem* p, const Geometry& geo, int obs_c0, unsigned terms,
                      const T* x, T* g, double* partials, int* nblocks, hipStream_t st);

// ---- forward tile kernel for sub-pixel shifts (kernels_spfwd.hip) ----
struct SpForwardPlan {
  void* d_frames = nullptr;  // per frame: integer offsets + blur (x) bilinear stencil
  int RLO = 0, CLO = 0, XR = 0, XC = 0;  // LDS window of a workgroup relative to its first LR row / cell
  bool ok = false;
  int RF0 = 0, NRF = 0, CF0 = 0, NCF = 0;  // union of the window and the workgroup's own HR block (rows / cells FOLD instances walk)
  bool can_fold = false;     // that union fits the kernel's load loop: the trial point can be formed (folded) here
};
// solver line search: the evaluation's point is xk + stp * d (d = dvec, scaled by the factors of `norms` when given:
// cg_norm.hpp); the forward kernel forms it as it loads its window and writes it to the evaluation's x.  xk == nullptr: none
struct SpFold { const void* xk = nullptr; const void* dvec = nullptr; double stp = 0.0; const double* norms = nullptr; };
bool spfwd_plan(srmap_problem* p, SpForwardPlan* sp);
void spfwd_release(SpForwardPlan* sp);
// out[k][c][h][w] = A_k x - y_k for all frames + cost partials (one per workgroup)
template <typename T>
int launch_forward_sp(srmap_problem* p, const Geometry& geo, const SpForwardPlan& sp, const T* x, const T* y,
                      int obs_C, int obs_c0, T* out, double* partials, int* nblocks, hipStream_t st,
                      const SpFold& fold = SpFold());

// ---- vector kernels for the solver (solver.hip) ----
int solve_impl(srmap_problem* p, srmap_comm* comm, const srmap_shard_desc* shard,
               const srmap_irls_options* o, const double* x0, double* x_out,
               srmap_solve_report* rep);

// conversions / staging
int ensure_staging(srmap_ctx* ctx);
int convert_upload(srmap_problem* p, const double* host, void* dev, size_t n,
                   hipStream_t st);
int convert_download(srmap_problem* p, const void* dev, double* host, size_t n,
                     hipStream_t st);

}  // namespace srmap
