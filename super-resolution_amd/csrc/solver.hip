// solver.hip -- IRLSMapSolver::Solve on the GPU (irls_map_solver.cpp:45-157,
// 192-265) with the nonlinear CG the reference obtains from ALGLIB 3.10.0
// (mincg, default settings: DY/HS hybrid beta, More'-Thuente line search,
// libs/alglib/src/optimization.cpp:17137-17880, alglibinternal.cpp:12313-12632),
// single-GPU or sharded over one rank per GPU.
//
// Every n-vector (iterate, gradient, directions, line-search base point) lives
// in HBM and is touched only by the kernels below.  The control flow (step
// selection, stopping rules) runs on the host, in double, in ALGLIB's order of
// operations, so that the trajectory follows the reference's up to reduction
// order.  Per CG iteration the host waits for the device 2 + nfev times: once
// for the direction's sums, once per trial point for (f, g.d), once for the beta
// dot products; the n-vector work is two fused passes:
//   k_direction      dn = -g + beta dk, max|dn|, dn.dn and g.dn -- from which the
//                    host derives linminnormalized's two scale factors, g.d and
//                    d.d (cg_norm.hpp): the normalised direction d = dn s1 s2 is
//                    not re-summed, and on the tile path not even stored
//   k_beta_dots      y = g - g_prev on the fly (the gradient buffers ping-pong,
//                    mincg's yk vector is never stored), g.g, g.y; their
//                    denominator y.dk = g.dk - g_prev.dk from sums already known
// plus the trial points x = xk + stp d: formed by the evaluation itself as it
// loads its window, from dk and the device-resident norms (tile kernel, un-sharded
// solves: no n-vector pass per trial point); elsewhere k_normalize stores d (and
// the first trial point) and k_axpy_out the later ones.  Each pass reduces its sums
// in the SAME launch: every block publishes its partials as write-through
// granules, the last block of the grid adds them in index order and hands the
// results (and the arrival tag) to the host -- no one-block second kernel.
//
// Sharding.  Reductions run over the elements a rank OWNS (row band or channel
// block; everything for frame shards) and are all-reduced through the
// communicator (sum, and max for the max-norm), so every rank takes the same
// decisions; x halos are refreshed before every evaluation (comm.hpp).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "cg_norm.hpp"
#include "comm.hpp"

namespace srmap {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}

constexpr int kRedBlocks = 1024;

// Which elements of an n-vector a rank owns: element range [e0, e1) (channel block) and, inside each H x W plane,
// rows [r0, r1) (row band).  on == 0: everything.
struct Owned {
  size_t e0, e1;
  int W, H, r0, r1;
  int on;
  __device__ __forceinline__ bool has(size_t i) const {
    if (!on) return true;
    if (i < e0 || i >= e1) return false;
    const int row = (int)((i / (size_t)W) % (size_t)H);
    return row >= r0 && row < r1;
  }
};

// ---- one-launch reductions -------------------------------------------------------------------------------
// A granule that has not been published yet holds this NaN pattern (both halves equal: hipMemsetD32 arms it).
constexpr unsigned kArm32 = 0x7FF9ABCDu;
constexpr unsigned long long kArm = ((unsigned long long)kArm32 << 32) | kArm32;
__device__ __forceinline__ unsigned long long ld_dev(const unsigned long long* q) {
  return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_dev(unsigned long long* q, unsigned long long v) {
  __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Where a pass leaves its reduced sums.  gran == nullptr: the two-launch scheme (block partials in `part`, k_finish
// follows; sharded solves, whose sums go through an all-reduce first).  Otherwise the last block of the grid reduces:
// out[0 .. rows) (device or host-mapped), the evaluation's cost forwarded from cost_src to out[rows], pub_n further
// device scalars copied to pub_dst (host-mapped), then the arrival tag behind a system-scope fence.
struct Fin {
  unsigned long long* gran;
  double* out;
  const double* cost_src;
  const double* pub_src;
  double* pub_dst;
  int pub_n;
  double* tag_slot;
  double tag;
  double* timeout_flag;   // sticky device word: a reduction gave up waiting for a block (srmap_solve reports it)
  double* timeout_host;   // the same event for the host at once (host-mapped word: wait_tag ends the solve on it)
};

// Block partials of up to 3 sums (row 0 a max when max0).  Two-launch scheme: part[k * gridDim.x + blockIdx.x].
// One-launch scheme: granules, and the grid's last block adds all of them -- per thread i = tid, tid + 256, ... in
// ascending order, then the wave and the four-wave combination of k_finish: the same additions in the same order as
// the two-launch scheme.  Returns true in the one thread that wrote out[] (it still owes fin_tag()).
__device__ __forceinline__ bool block_partials3(double s0, double s1, double s2, double* __restrict__ part, bool max0,
                                                int rows, const Fin& fin, double* tot = nullptr) {
  __shared__ double red[3][4];
  s0 = max0 ? wmax(s0) : wsum(s0);
  s1 = wsum(s1);
  s2 = wsum(s2);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { red[0][wid] = s0; red[1][wid] = s1; red[2][wid] = s2; }
  __syncthreads();
  const int nbk = gridDim.x;
  if (threadIdx.x < 3) {
    const double* r = red[threadIdx.x];
    const double v = (threadIdx.x == 0 && max0) ? fmax(fmax(r[0], r[1]), fmax(r[2], r[3])) : (r[0] + r[1]) + (r[2] + r[3]);
    if (fin.gran == nullptr) part[(size_t)threadIdx.x * nbk + blockIdx.x] = v;
    else if ((int)threadIdx.x < rows) st_dev(fin.gran + (size_t)threadIdx.x * nbk + blockIdx.x, (unsigned long long)__double_as_longlong(v));
  }
  if (fin.gran == nullptr || (int)blockIdx.x != nbk - 1) return false;
  __syncthreads();  // red[] is reused below
  double v0 = 0, v1 = 0, v2 = 0;
  bool timed_out = false;
  for (int i = threadIdx.x; i < nbk; i += 256) {
    unsigned long long a0 = 0, a1 = 0, a2 = 0;  // +0.0 for absent rows
    // bounded like the tile kernel's finisher (~2 s): a block that never publishes ends the pass with NaN sums (the
    // host's stopping rules then end the solve) instead of hanging the stream
    for (unsigned spins = 0;; ++spins) {
      if (rows > 0) a0 = ld_dev(fin.gran + i);
      if (rows > 1) a1 = ld_dev(fin.gran + (size_t)nbk + i);
      if (rows > 2) a2 = ld_dev(fin.gran + (size_t)2 * nbk + i);
      if (a0 != kArm && a1 != kArm && a2 != kArm) break;
      if (spins > (1u << 22)) { timed_out = true; break; }
      if (spins < 64) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(16);
    }
    if (timed_out) continue;   // NOT re-armed: a block that arrives late must not publish into a fresh slot
    if (rows > 0) st_dev(fin.gran + i, kArm);  // re-armed for the next pass
    if (rows > 1) st_dev(fin.gran + (size_t)nbk + i, kArm);
    if (rows > 2) st_dev(fin.gran + (size_t)2 * nbk + i, kArm);
    const double d0 = __longlong_as_double((long long)a0), d1 = __longlong_as_double((long long)a1), d2 = __longlong_as_double((long long)a2);
    v0 = max0 ? fmax(v0, d0) : v0 + d0;
    v1 += d1;
    v2 += d2;
  }
  v0 = max0 ? wmax(v0) : wsum(v0);
  v1 = wsum(v1);
  v2 = wsum(v2);
  // a time-out anywhere in the block makes EVERY row NaN (fmax would drop a NaN partial of the max row): the host's
  // stopping rules end the solve, srmap_solve reports SRMAP_EHIP (sticky word fin.timeout_flag) and re-initialises the granules
  const bool any_to = __syncthreads_or(timed_out ? 1 : 0) != 0;
  if (lane == 0) { red[0][wid] = v0; red[1][wid] = v1; red[2][wid] = v2; }
  __syncthreads();
  if (threadIdx.x != 0) return false;
  if (any_to) {
    const double qn = __builtin_nan("");
    red[0][0] = qn; red[1][0] = qn; red[2][0] = qn;
    if (fin.timeout_flag != nullptr) fin.timeout_flag[0] = 1.0;
    if (fin.timeout_host != nullptr) *(volatile double*)fin.timeout_host = 1.0;
  }
  const double t0 = max0 ? fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3]))
                         : (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  const double t1 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const double t2 = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
  if (rows > 0) fin.out[0] = t0;
  if (rows > 1) fin.out[1] = t1;
  if (rows > 2) fin.out[2] = t2;
  if (tot != nullptr) { tot[0] = t0; tot[1] = t1; tot[2] = t2; }  // the same sums for the finishing thread's own use
  if (fin.cost_src != nullptr) fin.out[rows] = fin.cost_src[0];
  for (int i = 0; i < fin.pub_n; ++i) fin.pub_dst[i] = fin.pub_src[i];
  return true;
}
__device__ __forceinline__ void fin_tag(const Fin& fin) {
  if (fin.tag_slot != nullptr) {
    __threadfence_system();
    *(volatile double*)fin.tag_slot = fin.tag;
  }
}

// Cache policy of the n-vector passes.  What an EVALUATION touches -- x, the observations, the IRLS weights, the
// direction d (its g.d), g -- and the line search's base point xk (read for every trial point) should survive in the
// 256 MiB Infinity Cache from one evaluation to the next: 201 MB at cfg2.  The vectors only the CG update itself
// streams (dn / dk, the previous gradient gp) are read and written NON-TEMPORAL so that they do not displace that set
// (profiles/r03_solve_trace.txt: the evaluation ran 51.7 us inside the solve against 39.9 us alone).
// V consecutive elements as one request (V * sizeof(T) <= 16 bytes, p aligned to it); STREAM = non-temporal
template <typename T, int V, bool STREAM>
__device__ __forceinline__ void ldv(const T* __restrict__ p, T (&out)[V]) {
  typedef T __attribute__((ext_vector_type(V))) VT;
  if (V == 1) { out[0] = STREAM ? __builtin_nontemporal_load(p) : *p; return; }
  const VT v = STREAM ? __builtin_nontemporal_load(reinterpret_cast<const VT*>(p)) : *reinterpret_cast<const VT*>(p);
#pragma unroll
  for (int q = 0; q < V; ++q) out[q] = v[q];
}
template <typename T, int V, bool STREAM>
__device__ __forceinline__ void stv(T* __restrict__ p, const T (&in)[V]) {
  typedef T __attribute__((ext_vector_type(V))) VT;
  if (V == 1) { if (STREAM) __builtin_nontemporal_store(in[0], p); else *p = in[0]; return; }
  VT v;
#pragma unroll
  for (int q = 0; q < V; ++q) v[q] = in[q];
  if (STREAM) __builtin_nontemporal_store(v, reinterpret_cast<VT*>(p)); else *reinterpret_cast<VT*>(p) = v;
}

// Every pass handles V consecutive elements per thread and step (V * sizeof(T) = 16 bytes per request when n is a
// multiple of V, else V = 1): the element-wise results are the same bits either way, the dot-product partials are
// summed in a different order (tests/test_gpu_parity.py: the PSNR bar of the ill-conditioned small cases follows the CPU
// reference path's own sensitivity to a last-bit perturbation).

// dn = -g + beta * dk ; sums: [0] max |dn| (owned), [1] dn.dn (owned), [2] g.dn (owned).  From these the host (and the
// kernels that need the normalised direction d = dn s1 s2) derive s1, s2, g.d = (g.dn s1) s2 and d.d = dn.dn s1^2 s2^2:
// the normalisation pass of rounds 1-4 (k_normalize_dots: five n-vector streams and a reduction per CG iteration, only
// to re-sum g.d and d.d over the stored d) is gone from every path; where d is needed as a vector a plain scaling pass
// (k_normalize) stores it.  norms_pub (host-mapped), when given, receives the three sums from the finishing thread
// ahead of the tag.  keep_dn: dn is read again by the evaluations (trial points formed from dn): stored with the
// default cache policy instead of non-temporal.
template <typename T, int V>
__global__ __launch_bounds__(256) void k_direction(T* __restrict__ dn, const T* __restrict__ g,
                                                  const T* __restrict__ dk, T beta, size_t n, Owned ow,
                                                  double* __restrict__ part, Fin fin, const double* __restrict__ beta_dev,
                                                  double* norms_pub, int keep_dn) {
  // beta_dev: the beta the preceding k_beta_dots left on the device (the host queues this pass without waiting for it)
  if (beta_dev != nullptr) beta = (T)beta_dev[0];
  double mx = 0, ss = 0, gd = 0;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V; i < n; i += (size_t)gridDim.x * 256 * V) {
    T gi[V], di[V], v[V];
    ldv<T, V, false>(g + i, gi);
    if (dk != nullptr) ldv<T, V, true>(dk + i, di);
#pragma unroll
    for (int q = 0; q < V; ++q) {
      v[q] = -gi[q];
      if (dk != nullptr) v[q] += beta * di[q];
    }
    if (keep_dn) stv<T, V, false>(dn + i, v); else stv<T, V, true>(dn + i, v);
#pragma unroll
    for (int q = 0; q < V; ++q)
      if (ow.has(i + q)) {
        mx = fmax(mx, fabs((double)v[q])); ss += (double)v[q] * (double)v[q]; gd += (double)gi[q] * (double)v[q];
      }
  }
  if (block_partials3(mx, ss, gd, part, true, 3, fin)) {
    if (norms_pub != nullptr) { norms_pub[0] = fin.out[0]; norms_pub[1] = fin.out[1]; norms_pub[2] = fin.out[2]; }
    fin_tag(fin);
  }
}

// Second stage: rows (<= 3) x nb partials -> out[rows] in fixed order; row 0 is a max when max0.  extra_src, when
// given, is one more device scalar (the cost of the evaluation) forwarded to out[rows].  When `tag_slot` is given
// (host-mapped memory) the kernel finally stores `tag` there behind a system-scope fence: the host polls that word
// instead of paying a stream synchronisation (tens of microseconds per wait on this runtime).
__global__ __launch_bounds__(256) void k_finish(const double* __restrict__ part, int nb, int rows, int max0,
                                               double* __restrict__ out, const double* __restrict__ extra_src,
                                               double* tag_slot, double tag) {
  __shared__ double red[3][4];
  double v0 = 0, v1 = 0, v2 = 0;
  // four partials per row and thread requested together (nb <= 1024: one round trip instead of four), added in the
  // same order as before
  constexpr int U = 4;
  for (int base = threadIdx.x; base < nb; base += 256 * U) {
    double a0[U], a1[U], a2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * 256;
      const bool in = i < nb;
      a0[u] = (in && rows > 0) ? part[i] : 0.0;
      a1[u] = (in && rows > 1) ? part[(size_t)nb + i] : 0.0;
      a2[u] = (in && rows > 2) ? part[(size_t)2 * nb + i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v0 = max0 ? fmax(v0, a0[u]) : v0 + a0[u];
      v1 += a1[u];
      v2 += a2[u];
    }
  }
  v0 = max0 ? wmax(v0) : wsum(v0);
  v1 = wsum(v1);
  v2 = wsum(v2);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { red[0][wid] = v0; red[1][wid] = v1; red[2][wid] = v2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (rows > 0) out[0] = max0 ? fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3]))
                                : (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    if (rows > 1) out[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    if (rows > 2) out[2] = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    if (extra_src != nullptr) out[rows] = extra_src[0];
    if (tag_slot != nullptr) {
      __threadfence_system();
      *(volatile double*)tag_slot = tag;
    }
  }
}

// dst[0..n) = src[0..n) (device scalars -> host-mapped memory), then the tag (see k_finish)
__global__ void k_publish(double* __restrict__ dst, const double* __restrict__ src, int n, double* tag_slot, double tag) {
  if (threadIdx.x == 0) {
    for (int i = 0; i < n; ++i) dst[i] = src[i];
    __threadfence_system();
    *(volatile double*)tag_slot = tag;
  }
}

// d = (dn * s1) * s2 stored as a vector, for the paths whose evaluations read the normalised direction from memory
// (sharded solves, the direct kernels, host-paced passes).  norms = device {max|dn|, dn.dn} (already all-reduced); every
// thread derives the same two factors.  When the first step of the line search is known before this pass (ALGLIB's
// lastgoodstep), its trial point x1 = xk + stp1 * d is written here as well: one pass over xk / x less per CG iteration
// than a separate k_axpy_out (same expression, same rounding).
template <typename T, int V>
__global__ __launch_bounds__(256) void k_normalize(T* __restrict__ d, const T* __restrict__ dn, const double* __restrict__ norms,
                                                  size_t n, const T* __restrict__ xk, T* __restrict__ x1, T stp1) {
  const double mx = norms[0], ss = norms[1];
  double s1, s2;
  norm_factors(mx, ss, s1, s2);
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V; i < n; i += (size_t)gridDim.x * 256 * V) {
    T dv[V], xv[V], v[V];
    ldv<T, V, true>(dn + i, dv);
    if (x1 != nullptr) ldv<T, V, false>(xk + i, xv);
#pragma unroll
    for (int q = 0; q < V; ++q) v[q] = norm_elem<T>(dv[q], mx, s1, s2);
    stv<T, V, false>(d + i, v);
    if (x1 != nullptr) {  // the line search's first trial point (k_axpy_out's expression)
      T xn[V];
#pragma unroll
      for (int q = 0; q < V; ++q) xn[q] = xv[q] + stp1 * v[q];
      stv<T, V, false>(x1 + i, xn);
    }
  }
}

// y = g - gp (mincg: yk = -g_k, then yk += g_{k+1}: the same rounding) ; sums: [0] g.g, [1] g.y   (the DY / HS betas,
// optimization.cpp:17700-17760).  Their denominator vv = y.dk is not summed here: y.dk = g.dk - gp.dk, and both terms are
// already known -- gp.dk is the g.dn the direction pass reduced, g.dk = (g.d) / (s1 s2) from the accepted trial
// evaluation's g.d (the line search bounds |g.d| by 0.3 |gp.d|: no cancellation) -- so the pass reads two vectors
// instead of three (dk is not touched).  vv comes as an argument.
template <typename T, int V>
__global__ __launch_bounds__(256) void k_beta_dots(const T* __restrict__ gp, const T* __restrict__ g,
                                                  size_t n, Owned ow, double* __restrict__ part, Fin fin,
                                                  double* __restrict__ beta_dst, int restart, double vv,
                                                  const T* __restrict__ dk_check) {
  // dk_check (host-paced passes only): the denominator y.dk summed directly as well, row [2] -- the self-check of the
  // derived vv (srmap_problem_selfcheck); the betas still use the derived one, so both pacing modes stay bit-equal
  double b = 0, c = 0, e = 0;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V; i < n; i += (size_t)gridDim.x * 256 * V) {
    T gv[V], pv[V], kv[V];
    ldv<T, V, false>(g + i, gv);
    ldv<T, V, true>(gp + i, pv);
    if (dk_check != nullptr) ldv<T, V, true>(dk_check + i, kv);
#pragma unroll
    for (int q = 0; q < V; ++q) {
      if (!ow.has(i + q)) continue;
      const T y = -pv[q] + gv[q];
      b += (double)gv[q] * (double)gv[q]; c += (double)gv[q] * (double)y;
      if (dk_check != nullptr) e += (double)y * (double)kv[q];
    }
  }
  double tot[3];
  if (block_partials3(b, c, e, part, false, dk_check != nullptr ? 3 : 2, fin, tot)) {
    if (beta_dst != nullptr) {
      // betak = max(0, min(betady, betahs)) exactly as run_cg forms it on the host (same IEEE divisions and compares):
      // the direction pass queued behind this one reads it, the host never has to answer in between
      const double bdy = tot[0] / vv, bhs = tot[1] / vv;
      const double bm = bdy < bhs ? bdy : bhs;
      double bk = 0.0 > bm ? 0.0 : bm;
      if (restart) bk = 0.0;
      beta_dst[0] = bk;
    }
    fin_tag(fin);
  }
}

// partial of a.b over the owned elements: [0]
template <typename T, int V>
__global__ __launch_bounds__(256) void k_dot(const T* __restrict__ a, const T* __restrict__ b, size_t n, Owned ow,
                                            double* __restrict__ part, Fin fin) {
  double s = 0;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V; i < n; i += (size_t)gridDim.x * 256 * V) {
    T av[V], bv[V];
    ldv<T, V, false>(a + i, av);
    ldv<T, V, false>(b + i, bv);
#pragma unroll
    for (int q = 0; q < V; ++q)
      if (ow.has(i + q)) s += (double)av[q] * (double)bv[q];
  }
  if (block_partials3(s, 0.0, 0.0, part, false, 1, fin)) fin_tag(fin);
}

// dst = a + alpha * b
template <typename T>
__global__ void k_axpy_out(T* __restrict__ dst, const T* __restrict__ a, const T* __restrict__ b, T alpha, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = a[i] + alpha * b[i];
}
// the same, four elements per thread (16 / 32-byte requests; hipMalloc'ed vectors are aligned): element by element the
// same expression, so the same bits
template <typename T>
__global__ __launch_bounds__(256) void k_axpy_out4(T* __restrict__ dst, const T* __restrict__ a, const T* __restrict__ b, T alpha,
                                                  size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  T va[4], vb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { va[q] = a[4 * i + q]; vb[q] = b[4 * i + q]; }
#pragma unroll
  for (int q = 0; q < 4; ++q) dst[4 * i + q] = va[q] + alpha * vb[q];
}
template <typename T>
__global__ void k_fill(T* __restrict__ d, T v, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = v;
}

// ---------------------------------------------------------------------------------------------------------
// sharded evaluation (comm.hpp)
int shard_exchange_x(srmap_problem* p, srmap_comm* c, const srmap_shard_desc* sd, void* x_dev, hipStream_t st) {
  if (!c || !sd || comm_world(c) <= 1) return SRMAP_OK;
  const Geometry& g = p->geo;
  const size_t N = (size_t)g.W * g.H, es = p->elem();
  const int rank = comm_rank(c), world = comm_world(c);
  const int up = rank > 0 ? rank - 1 : -1, down = rank + 1 < world ? rank + 1 : -1;
  char* x = (char*)x_dev;
  if (sd->mode == SRMAP_SHARD_ROWS) {
    // a frame whose warpAffine y coordinate sits on a 1/32-px rounding tie carries a per-row table built from the row
    // index of THIS problem (srmap_api.hip make_warp): a band problem would evaluate the tie at its local rows, not
    // the joint image's
    if (!p->d_ytabs.empty())
      return set_error(p->ctx, SRMAP_EUNSUPPORTED, "row shard: a sub-pixel shift on a 1/32-px rounding tie needs the joint image's row index; shard such problems by frames or channels");
    const int hu = sd->own_row0, hd = g.H - sd->own_row1;  // my halo rows above / below
    if ((up >= 0 && hu == 0) || (down >= 0 && hd == 0) || sd->send_down_rows > sd->own_row1 - sd->own_row0 ||
        sd->send_up_rows > sd->own_row1 - sd->own_row0)
      return set_error(p->ctx, SRMAP_EINVAL, "row shard: halo description inconsistent");
    if ((up >= 0 && sd->send_up_rows <= 0) || (down >= 0 && sd->send_down_rows <= 0))
      return set_error(p->ctx, SRMAP_EINVAL, "row shard: a neighbour exists but no rows are sent to it (it would wait for them)");
    std::vector<const void*> sa(g.C), sb(g.C);
    std::vector<void*> ra(g.C), rb(g.C);
    for (int ch = 0; ch < g.C; ++ch) {
      // downward traffic: my last owned rows -> lower neighbour's top halo; my top halo <- upper neighbour
      sa[ch] = x + ((size_t)ch * N + (size_t)(sd->own_row1 - sd->send_down_rows) * g.W) * es;
      ra[ch] = x + ((size_t)ch * N) * es;
      // upward traffic: my first owned rows -> upper neighbour's bottom halo; my bottom halo <- lower neighbour
      sb[ch] = x + ((size_t)ch * N + (size_t)sd->own_row0 * g.W) * es;
      rb[ch] = x + ((size_t)ch * N + (size_t)sd->own_row1 * g.W) * es;
    }
    // both directions in ONE group (one launch on the stream)
    return comm_exchange2(c, sa.data(), ra.data(), (size_t)sd->send_down_rows * g.W, (size_t)hu * g.W, sb.data(), rb.data(),
                          (size_t)sd->send_up_rows * g.W, (size_t)hd * g.W, up, down, g.C, p->dtype, st);
  }
  if (sd->mode == SRMAP_SHARD_CHANNELS || sd->mode == SRMAP_SHARD_GRID) {
    const bool lo = sd->own_ch0 > 0, hi = sd->own_ch1 < g.C;  // halo planes present (3-D TV coupling)
    if (!lo && !hi) return SRMAP_OK;
    // GRID: the channel neighbours are the ranks of the same frame group in the adjacent channel blocks
    const int stride = sd->mode == SRMAP_SHARD_GRID ? (sd->frame_groups > 0 ? sd->frame_groups : 1) : 1;
    const int cup = rank - stride >= 0 ? rank - stride : -1, cdown = rank + stride < world ? rank + stride : -1;
    // downward: my last owned plane -> lower neighbour's low halo plane; my low halo <- upper neighbour
    const void* s1 = x + (size_t)(sd->own_ch1 - 1) * N * es;
    void* r1 = x + (size_t)(sd->own_ch0 - 1) * N * es;
    // upward: my first owned plane -> upper neighbour's high halo plane; my high halo <- lower neighbour
    const void* s2 = x + (size_t)sd->own_ch0 * N * es;
    void* r2 = x + (size_t)sd->own_ch1 * N * es;
    return comm_exchange2(c, &s1, &r1, hi ? N : 0, lo ? N : 0, &s2, &r2, lo ? N : 0, hi ? N : 0, lo ? cup : -1, hi ? cdown : -1, 1,
                          p->dtype, st);
  }
  return SRMAP_OK;
}

int shard_eval(srmap_problem* p, srmap_comm* c, const srmap_shard_desc* sd, unsigned terms, void* x_dev, void* g_dev,
               hipStream_t st) {
  const int mode = (c && sd && comm_world(c) > 1) ? sd->mode : SRMAP_SHARD_NONE;
  if (mode == SRMAP_SHARD_NONE) return srmap_eval_device(p, terms, x_dev, g_dev, nullptr, st);
  const size_t N = (size_t)p->geo.W * p->geo.H, es = p->elem();
  if (mode == SRMAP_SHARD_ROWS) {
    // The halo rows of x travel on the communicator's side stream while the evaluation's stream runs the tile rows
    // that read none of them; the boundary tile rows wait for the event (kernels_ztile.hip launch_z; paths without
    // that split exchange first).  x is ready when `st` reaches this point; the next exchange cannot start before
    // this evaluation (which reads the halos) is behind the next ev_x.
    // Overlap only where it is both enabled on the communicator and SAFE: the caller's halo must be at least the tile
    // kernel's reach (x rows 2 above / 3 below a tile row: blur transpose + regulariser window), otherwise an
    // "interior" tile row would read a row the exchange is still writing.
    const int hu = sd->own_row0, hd = p->geo.H - sd->own_row1;
    constexpr int kReach = 4;
    const bool overlap = comm_overlap(c) && ztile_overlaps_halo(p) && (hu == 0 || hu >= kReach) && (hd == 0 || hd >= kReach);
    if (!overlap) {
      int rc = shard_exchange_x(p, c, sd, x_dev, st);
      if (rc) return rc;
      return srmap_eval_device(p, terms, x_dev, g_dev, nullptr, st);
    }
    struct Hook { srmap_problem* p; srmap_comm* c; const srmap_shard_desc* sd; void* x; hipStream_t side; hipEvent_t ev; bool called; };
    hipStream_t side; hipEvent_t ev_x, ev_halo;
    int rc = comm_side(c, &side, &ev_x, &ev_halo);
    if (rc) return rc;
    SRMAP_HIP(p->ctx, hipEventRecord(ev_x, st));
    SRMAP_HIP(p->ctx, hipStreamWaitEvent(side, ev_x, 0));
    Hook h{p, c, sd, x_dev, side, ev_halo, false};
    p->ov_hook = [](void* a) -> int {
      Hook* k = static_cast<Hook*>(a);
      k->called = true;
      int r = shard_exchange_x(k->p, k->c, k->sd, k->x, k->side);
      if (r) return r;
      SRMAP_HIP(k->p->ctx, hipEventRecord(k->ev, k->side));
      return SRMAP_OK;
    };
    p->ov_arg = &h;
    p->ov_event = ev_halo;
    p->ov_top = hu;
    p->ov_bot = hd;
    rc = srmap_eval_device(p, terms, x_dev, g_dev, nullptr, st);  // cost rows were set on the problem
    p->ov_hook = nullptr; p->ov_arg = nullptr; p->ov_event = nullptr;
    // An evaluation that failed before it reached the hook has not posted this rank's half of the exchange: post it
    // now, so that the neighbours' receives complete and they see an error code instead of a hang.
    if (!h.called) {
      const int rx = shard_exchange_x(p, c, sd, x_dev, side);
      if (rc == SRMAP_OK) rc = rx;
    }
    return rc;
  }
  int rc = shard_exchange_x(p, c, sd, x_dev, st);
  if (rc) return rc;
  if (mode == SRMAP_SHARD_FRAMES) {
    // Every rank adds its frames' data term.  The regulariser is evaluated once over the ranks: split by row band
    // (whole tile rows, balanced) when the tile kernel alone produces it -- at cfg2-class mixes it is more than half of
    // the arithmetic, so leaving it to one rank would make that rank the critical path -- else on reg_rank.
    const int rank = comm_rank(c), world = comm_world(c);
    unsigned t = terms;
    // The split is a COLLECTIVE decision: a rank whose own frame subset has no tile plan (a shift on a 1/32-px rounding
    // tie, a per-rank SRMAP_IMPL_DIRECT, ...) cannot evaluate a band, and if it went its own way the regulariser would
    // be counted twice or not at all.  The ranks agree once (minimum of their flags over the communicator; cached on
    // the problem until its plan generation -- bumped by every re-plan and every srmap_problem_set_impl --, term set or
    // communicator changes); any rank that cannot band-split sends everybody to reg_rank.  The agreement is itself a
    // collective: under frame sharding srmap_problem_set_impl and the regulariser calls are COLLECTIVE too (every rank
    // makes them in the same order between the same evaluations; include/srmap.h), or one rank would enter it alone.
    const bool mine = ztile_reg_band_ok(p, terms);
    if (p->band_comm != (const void*)c || p->band_terms != terms || p->band_gen != p->plan_gen) {
      double flag = mine ? 0.0 : 1.0;  // max over the ranks of "I cannot" == 0  <=>  every rank can
      SRMAP_HIP(p->ctx, hipMemcpyAsync(p->d_cost + 7, &flag, sizeof(double), hipMemcpyHostToDevice, st));
      rc = comm_allreduce(c, p->d_cost + 7, 1, SRMAP_F64, 1, st);
      if (rc) return rc;
      SRMAP_HIP(p->ctx, hipMemcpyAsync(&flag, p->d_cost + 7, sizeof(double), hipMemcpyDeviceToHost, st));
      SRMAP_HIP(p->ctx, hipStreamSynchronize(st));
      p->band_all = flag == 0.0;
      p->band_comm = c; p->band_terms = terms; p->band_gen = p->plan_gen;
    }
    const bool band = mine && p->band_all;
    if (band) {
      const int tiles = (p->geo.H + 7) / 8, per = (tiles + world - 1) / world;
      p->geo.rr0 = std::min(p->geo.H, rank * per * 8);
      p->geo.rr1 = std::min(p->geo.H, (rank + 1) * per * 8);
    } else if (rank != sd->reg_rank) {
      t = terms & SRMAP_TERM_DATA;
    }
    if (t == 0) {
      SRMAP_HIP(p->ctx, hipMemsetAsync(p->d_cost, 0, sizeof(double), st));
      if (g_dev) SRMAP_HIP(p->ctx, hipMemsetAsync(g_dev, 0, p->hr_count() * es, st));
    } else {
      rc = srmap_eval_device(p, t, x_dev, g_dev, nullptr, st);
    }
    p->geo.rr0 = 0; p->geo.rr1 = p->geo.H;
    if (rc) return rc;
    // the north-star's gradient all-reduce, with the cost in the same group (one launch)
    return comm_allreduce_grad_cost(c, g_dev, g_dev ? p->hr_count() : 0, p->dtype, p->d_cost, st);
  }
  if (mode == SRMAP_SHARD_CHANNELS || mode == SRMAP_SHARD_GRID) {
    const int fgs = (mode == SRMAP_SHARD_GRID && sd->frame_groups > 1) ? sd->frame_groups : 1;
    const int fg = comm_rank(c) % fgs;
    // GRID: the regulariser terms of a channel block are evaluated once, by its frame group 0
    const unsigned t = (fg == 0) ? terms : (terms & SRMAP_TERM_DATA);
    const int saved_c0 = p->view_c0, saved_C = p->view_C;
    const bool saved_cp = p->view_coupled;
    p->view_c0 = sd->own_ch0; p->view_C = sd->own_ch1 - sd->own_ch0; p->view_coupled = true;
    char* gown = g_dev ? (char*)g_dev + (size_t)sd->own_ch0 * N * es : nullptr;
    if (t == 0) {
      rc = SRMAP_OK;
      SRMAP_HIP(p->ctx, hipMemsetAsync(p->d_cost, 0, sizeof(double), st));
      if (gown) SRMAP_HIP(p->ctx, hipMemsetAsync(gown, 0, (size_t)p->view_C * N * es, st));
    } else {
      rc = srmap_eval_device(p, t, (char*)x_dev + (size_t)sd->own_ch0 * N * es, gown, nullptr, st);
    }
    const size_t cnt = (size_t)p->view_C * N;
    p->view_c0 = saved_c0; p->view_C = saved_C; p->view_coupled = saved_cp;
    if (rc) return rc;
    if (fgs > 1 && gown) {  // sum of the frame groups' data-term gradients of this channel block
      if (!sd->frame_comm) return set_error(p->ctx, SRMAP_EINVAL, "grid shard: frame_comm missing");
      rc = comm_allreduce(sd->frame_comm, gown, cnt, p->dtype, 0, st);
    }
    return rc;
  }
  return srmap_eval_device(p, terms, x_dev, g_dev, nullptr, st);  // rows: cost rows were set on the problem
}

// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct DeviceCG {
  srmap_problem* p;
  hipStream_t st;
  size_t n;
  srmap_comm* comm = nullptr;
  const srmap_shard_desc* shard = nullptr;
  Owned ow{};
  bool reduce_scalars = false;  // row / channel shards: the owned-element sums are all-reduced
  bool published = false;       // the last evaluation's finish kernel published {f, g.d} + tag (fetch_f_gd just waits)
  // chained passes (run_cg): launches whose inputs are already on the device are queued without waiting for the host.
  // srmap_irls_options::host_paced_passes turns this off (every pass then waits for the host's answer, as up to
  // round 3): same arithmetic, same results bit for bit -- tests/test_gpu_solve_parity.py compares the two.  Chaining
  // applies to un-sharded solves only: under frame sharding a speculative first-trial evaluation would queue a
  // gradient + cost all-reduce that every rank has to match before the host has decided to keep it.
  bool chain_enabled = true;
  bool chained() const { return fused() && chain_enabled && (shard == nullptr || comm == nullptr); }
  // x, g: current point and gradient; xk/dk: accepted point and direction; dn: next direction;
  // d: normalised direction (stored only where evaluations read it from memory: !foldable); gp: the gradient at xk
  // while the line search writes its trial gradients to g (the two
  // buffers swap; mincg's yk = g_{k+1} - g_k is formed on the fly).  The line-search base is xk itself.
  T *x = nullptr, *g = nullptr, *xk = nullptr, *dk = nullptr, *dn = nullptr, *d = nullptr, *gp = nullptr;
  double* part = nullptr;   // [3][kRedBlocks] block partials (two-launch reductions: sharded solves)
  unsigned long long* gran = nullptr;  // [3][kRedBlocks] granules of the one-launch reductions (armed)
  double* dscal = nullptr;  // device scalars: [0..3] reduction results, [4..6] the direction's sums {max|dn|, dn.dn, g.dn}, [8] beta, [15] time-out flag
  double* hs = nullptr;     // host-mapped pinned scalars (ctx->h_scal): pass results [0..3], the direction's sums [8..10], arrival tag [15]
  double tag = 0;           // last tag handed to a publishing kernel
  int evaluations = 0;
  double wait_seconds = 0;  // host time spent in wait_tag
  int waits = 0;

  // Wait until the kernel that was given tag `want` (default: the last one handed out) has published its results (see
  // k_finish): poll the host-mapped word, fall back to a stream synchronisation after ~2 s (also surfaces asynchronous
  // errors).  Tags only grow and the publishing kernels of one stream finish in order, so "arrived" is slot >= want:
  // a later kernel queued behind the awaited one (chained passes, run_cg) may already have stored its own tag.
  int wait_tag(double want = -1.0) {
    if (want < 0) want = tag;
    volatile double* slot = hs + 15;
    const auto t0 = std::chrono::steady_clock::now();
    struct Acc { DeviceCG* c; std::chrono::steady_clock::time_point t; ~Acc() {
      c->wait_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); c->waits++; } } acc{this, t0};
    unsigned spins = 0;
    volatile double* to = hs + 13;   // a device-side reduction of this solve timed out: its sums are NaN, its granules not re-armed
    while (!(*slot >= want)) {
      if (*to != 0.0) return set_error(p->ctx, SRMAP_EHIP, "solver: a device-side reduction timed out waiting for a workgroup");
      if ((++spins & 0x3ff) == 0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
        SRMAP_HIP(p->ctx, hipStreamSynchronize(st));
        if (!(*slot >= want)) return set_error(p->ctx, SRMAP_EHIP, "solver: device results did not arrive");
        break;
      }
    }
    return SRMAP_OK;
  }

  static constexpr int kVec = 16 / (int)sizeof(T);  // elements per 16-byte request of the n-vector passes
  bool vec() const { return n % kVec == 0; }           // (the vectors are hipMalloc'ed: aligned)
  unsigned blocks() const { return (unsigned)((n + 255) / 256); }
  int nb() const { size_t b = (n + 255) / 256; return (int)(b < (size_t)kRedBlocks ? b : kRedBlocks); }

  int alloc() {
    T** v[] = {&x, &g, &xk, &dk, &dn, &d, &gp};
    for (T** q : v) SRMAP_HIP(p->ctx, hipMalloc((void**)q, n * sizeof(T)));
    SRMAP_HIP(p->ctx, hipMalloc((void**)&part, sizeof(double) * 3 * kRedBlocks));
    SRMAP_HIP(p->ctx, hipMalloc((void**)&dscal, sizeof(double) * 16));
    SRMAP_HIP(p->ctx, hipMemsetAsync(dscal, 0, sizeof(double) * 16, st));
    SRMAP_HIP(p->ctx, hipMalloc((void**)&gran, sizeof(double) * 3 * kRedBlocks));
    SRMAP_HIP(p->ctx, hipMemsetD32Async((hipDeviceptr_t)gran, (int)kArm32, 2 * 3 * kRedBlocks, st));
    // sharded evaluations write only the owned part of g: the vector kernels run over all n elements, so everything
    // they combine starts defined (the halo values never enter a reduction, and x halos are re-exchanged)
    T* z[] = {g, dn, d, dk, gp};
    for (T* q : z) SRMAP_HIP(p->ctx, hipMemsetAsync(q, 0, n * sizeof(T), st));
    int rc = ensure_staging(p->ctx);
    if (rc) return rc;
    hs = p->ctx->h_scal;
    SRMAP_HIP(p->ctx, hipStreamSynchronize(st));
    hs[13] = 0.0;  // time-out word of this solve (Fin::timeout_host, ZArgs::to_host)
    tag = hs[15];  // tags keep increasing across solves of one context: a stale word can never match
    return SRMAP_OK;
  }
  void release() {
    T* v[] = {x, g, xk, dk, dn, d, gp};
    for (T* q : v) if (q) (void)hipFree(q);
    if (part) (void)hipFree(part);
    if (gran) (void)hipFree(gran);
    if (dscal) (void)hipFree(dscal);
  }
  int copy(T* dst, const T* src) {
    SRMAP_HIP(p->ctx, hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToDevice, st));
    return SRMAP_OK;
  }
  // One-launch reductions (struct Fin) whenever the sums need no all-reduce.
  bool fused() const { return !reduce_scalars; }
  // the pass about to be launched reduces to the host: out = hs[0 .. rows) (+ the cost at hs[rows]), then the tag
  Fin fin_host(bool with_cost, const double* pub_src = nullptr, double* pub_dst = nullptr, int pub_n = 0,
               double* out = nullptr) {
    Fin f{};
    if (fused()) {
      tag += 1.0;
      f.gran = gran; f.out = out ? out : hs; f.cost_src = with_cost ? (const double*)p->d_cost : nullptr;
      f.timeout_flag = dscal + 15; f.timeout_host = hs + 13;
      f.pub_src = pub_src; f.pub_dst = pub_dst; f.pub_n = pub_n;
      f.tag_slot = hs + 15; f.tag = tag;
    }
    return f;
  }
  // Bring the sums of the pass just launched to the host: out[0..rows) (+ out[rows] = cost).  One wait.  Two-launch
  // scheme: reduces the `rows` partial rows here (k_finish), all-reduces them, publishes.
  int finish(int rows, bool max0, bool with_cost, double* out, int extra_n = 0) {
    const int cnt = rows + (with_cost ? 1 : 0);
    if (!fused()) {
      tag += 1.0;
      hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, st, part, nb(), rows, max0 ? 1 : 0, dscal,
                         with_cost ? (const double*)p->d_cost : (const double*)nullptr, (double*)nullptr, 0.0);
      int rc = SRMAP_OK;
      if (max0) {
        rc = comm_allreduce(comm, dscal, 1, SRMAP_F64, 1, st);
        if (rc) return rc;
        if (cnt > 1) rc = comm_allreduce(comm, dscal + 1, (size_t)cnt - 1, SRMAP_F64, 0, st);
      } else {
        rc = comm_allreduce(comm, dscal, (size_t)cnt, SRMAP_F64, 0, st);
      }
      if (rc) return rc;
      if (extra_n > 0)
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, hs + 8, (const double*)(dscal + 4), extra_n, (double*)(hs + 14), 0.0);
      hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, hs, (const double*)dscal, cnt, hs + 15, tag);
    }
    SRMAP_HIP(p->ctx, hipGetLastError());
    int rc = wait_tag();
    if (rc) return rc;
    for (int i = 0; i < cnt; ++i) out[i] = hs[i];
    return SRMAP_OK;
  }
  // objective at x: g <- gradient; the cost stays on the device (finish(with_cost) fetches it).  With a direction
  // the tile kernel may produce g.d in the same pass (p->gd_valid; not under frame sharding, where the local
  // gradient is only a partial sum).
  // the line search may hand its trial point to the evaluation as (xk, stp): un-sharded solves on the tile kernel's
  // g.d instance (ztile_can_fold); decided once per CG run
  bool foldable = false;
  bool fold_enabled = true;   // srmap_irls_options::host_paced_passes also forms every trial point by its own pass (the A/B of the fold)
  int evaluate(const T* dir = nullptr, T* at = nullptr, const T* fold_xk = nullptr, double fold_stp = 0.0) {  // at: the point (default x)
    evaluations++;
    const int mode = (comm && shard && comm_world(comm) > 1) ? shard->mode : SRMAP_SHARD_NONE;
    p->eval_dvec = (mode == SRMAP_SHARD_FRAMES || mode == SRMAP_SHARD_CHANNELS || mode == SRMAP_SHARD_GRID) ? nullptr : dir;
    p->gd_valid = false;
    p->eval_published = false;
    // without a scalar all-reduce the evaluation's finish kernel can publish {f, g.d} and the arrival tag itself
    p->eval_pub = (!reduce_scalars && p->eval_dvec != nullptr) ? hs : nullptr;
    p->eval_pub_tag_slot = hs + 15;
    p->eval_pub_tag = tag + 1.0;
    p->eval_timeout_host = hs + 13;
    // fold: `dir` is the UNNORMALISED direction dk; the kernel scales it by the factors it derives from the norms the
    // direction pass left at dscal[4..5] (norm_factors / norm_elem: the bits of the stored d)
    p->eval_fold_xk = (fold_xk != nullptr && p->eval_dvec != nullptr) ? fold_xk : nullptr;
    p->eval_fold_stp = fold_stp;
    p->eval_fold_norms = p->eval_fold_xk != nullptr ? (const double*)(dscal + 4) : nullptr;
    const int rc = shard_eval(p, comm, shard, SRMAP_TERM_ALL, at ? at : x, g, st);
    p->eval_fold_xk = nullptr;
    p->eval_fold_norms = nullptr;
    p->eval_dvec = nullptr;
    p->eval_pub = nullptr;
    p->eval_timeout_host = nullptr;
    published = p->eval_published;
    if (published) tag += 1.0;
    return rc;
  }
  // The evaluation queued ahead of the host's decision (run_cg: the first trial point behind the normalisation pass)
  // turned out not to be wanted: it is not one of the solve's evaluations and nobody fetches its sums.  Its tag, if it
  // publishes one, is simply passed over (wait_tag compares with >=).
  void discard_speculative() {
    evaluations--;
    published = false;
  }
  // f and g.d of the evaluation just made, with one wait: out[0] = g.d, out[1] = f
  int fetch_f_gd(double* out) {
    if (!p->gd_valid) {
      if (vec()) hipLaunchKernelGGL((k_dot<T, kVec>), dim3(nb()), dim3(256), 0, st, (const T*)g, (const T*)d, n, ow, part, fin_host(true));
      else hipLaunchKernelGGL((k_dot<T, 1>), dim3(nb()), dim3(256), 0, st, (const T*)g, (const T*)d, n, ow, part, fin_host(true));
      return finish(1, false, true, out);
    }
    if (published) {  // the evaluation's own finish kernel carries the tag
      published = false;
      const int rcw = wait_tag();
      if (rcw) return rcw;
      out[0] = hs[1];
      out[1] = hs[0];
      return SRMAP_OK;
    }
    tag += 1.0;
    if (!reduce_scalars) {
      hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, hs, (const double*)p->d_cost, 2, hs + 15, tag);
    } else {
      SRMAP_HIP(p->ctx, hipMemcpyAsync(dscal, p->d_cost, 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
      int rc = comm_allreduce(comm, dscal, 2, SRMAP_F64, 0, st);
      if (rc) return rc;
      hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, hs, (const double*)dscal, 2, hs + 15, tag);
    }
    SRMAP_HIP(p->ctx, hipGetLastError());
    int rc = wait_tag();
    if (rc) return rc;
    out[0] = hs[1];
    out[1] = hs[0];
    return SRMAP_OK;
  }
  // dn = -g + beta dk (dk may be null); {max|dn|, dn.dn, g.dn} -> dscal[4..6] (device, all-reduced) and, under a tag of
  // the pass's own (dir_tag: wait_dir), hs[8..10].  publish_cost: the first pass of a CG run also hands f to the host
  // (one-launch scheme: hs[0] with the same tag; two-launch scheme: the caller's finish(0, ..., 3) publishes all four).
  // beta_dev: beta comes from the device scalar the k_beta_dots queued just before left there (one-launch scheme only)
  double dir_tag = 0;
  int direction(const T* dk_or_null, double beta, bool publish_cost = false, const double* beta_dev = nullptr) {
    Fin f{};
    if (fused()) {
      tag += 1.0;
      dir_tag = tag;
      f.gran = gran; f.out = dscal + 4; f.timeout_flag = dscal + 15; f.timeout_host = hs + 13;
      if (publish_cost) { f.pub_src = (const double*)p->d_cost; f.pub_dst = hs; f.pub_n = 1; }  // hs[0] = f
      f.tag_slot = hs + 15; f.tag = tag;
    }
    double* npub = fused() ? hs + 8 : (double*)nullptr;
    const int keep = foldable ? 1 : 0;
    if (vec()) hipLaunchKernelGGL((k_direction<T, kVec>), dim3(nb()), dim3(256), 0, st, dn, (const T*)g, dk_or_null, (T)beta, n, ow, part, f, beta_dev, npub, keep);
    else hipLaunchKernelGGL((k_direction<T, 1>), dim3(nb()), dim3(256), 0, st, dn, (const T*)g, dk_or_null, (T)beta, n, ow, part, f, beta_dev, npub, keep);
    if (!fused()) {
      hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, st, part, nb(), 3, 1, dscal + 4, (const double*)nullptr,
                         (double*)nullptr, 0.0);
      int rc = comm_allreduce(comm, dscal + 4, 1, SRMAP_F64, 1, st);
      if (rc) return rc;
      rc = comm_allreduce(comm, dscal + 5, 2, SRMAP_F64, 0, st);
      if (rc) return rc;
      if (!publish_cost) {
        tag += 1.0;
        dir_tag = tag;
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, hs + 8, (const double*)(dscal + 4), 3, hs + 15, tag);
      }
    }
    SRMAP_HIP(p->ctx, hipGetLastError());
    return SRMAP_OK;
  }
  // the sums of the last direction pass on the host: {max|dn|, dn.dn, g.dn}
  int wait_dir(double* mx, double* ss, double* gdn) {
    const int rc = wait_tag(dir_tag);
    if (rc) return rc;
    *mx = hs[8]; *ss = hs[9]; *gdn = hs[10];
    return SRMAP_OK;
  }
  // d = dk s1 s2 stored as a vector (+ the first trial point x = xk + stp1 d when stp1 != 0): the paths whose evaluations
  // read the normalised direction from memory
  int normalize(double stp1) {
    if (vec())
      hipLaunchKernelGGL((k_normalize<T, kVec>), dim3(nb()), dim3(256), 0, st, d, (const T*)dk, (const double*)(dscal + 4), n,
                         (const T*)xk, stp1 != 0.0 ? x : (T*)nullptr, (T)stp1);
    else
      hipLaunchKernelGGL((k_normalize<T, 1>), dim3(nb()), dim3(256), 0, st, d, (const T*)dk, (const double*)(dscal + 4), n,
                         (const T*)xk, stp1 != 0.0 ? x : (T*)nullptr, (T)stp1);
    SRMAP_HIP(p->ctx, hipGetLastError());
    return SRMAP_OK;
  }
};

static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double dmin(double a, double b) { return a < b ? a : b; }

// More'-Thuente safeguarded step (MINPACK-2 dcstep; ALGLIB linmin_mcstep,
// alglibinternal.cpp:12972-13232).
struct Bracket { double stx, fx, dx, sty, fy, dy; };

static double cubic_gamma(double theta, double da, double db, bool clamp0) {
  const double s = dmax(std::fabs(theta), dmax(std::fabs(da), std::fabs(db)));
  double t = (theta / s) * (theta / s) - da / s * (db / s);
  if (clamp0) t = dmax(0.0, t);
  return s * std::sqrt(t);
}

static void mt_step(Bracket* b, double* stp, double fp, double dp, bool* brackt, double stmin,
                    double stmax, int* info) {
  *info = 0;
  if ((*brackt && (*stp <= dmin(b->stx, b->sty) || *stp >= dmax(b->stx, b->sty))) ||
      b->dx * (*stp - b->stx) >= 0 || stmax < stmin)
    return;
  const double sgnd = dp * (b->dx / std::fabs(b->dx));
  bool bound;
  double stpf;
  if (fp > b->fx) {
    *info = 1; bound = true;
    const double theta = 3 * (b->fx - fp) / (*stp - b->stx) + b->dx + dp;
    double gamma = cubic_gamma(theta, b->dx, dp, false);
    if (*stp < b->stx) gamma = -gamma;
    const double pp = gamma - b->dx + theta, q = gamma - b->dx + gamma + dp, r = pp / q;
    const double stpc = b->stx + r * (*stp - b->stx);
    const double stpq = b->stx + b->dx / ((b->fx - fp) / (*stp - b->stx) + b->dx) / 2 * (*stp - b->stx);
    stpf = std::fabs(stpc - b->stx) < std::fabs(stpq - b->stx) ? stpc : stpc + (stpq - stpc) / 2;
    *brackt = true;
  } else if (sgnd < 0) {
    *info = 2; bound = false;
    const double theta = 3 * (b->fx - fp) / (*stp - b->stx) + b->dx + dp;
    double gamma = cubic_gamma(theta, b->dx, dp, false);
    if (*stp > b->stx) gamma = -gamma;
    const double pp = gamma - dp + theta, q = gamma - dp + gamma + b->dx, r = pp / q;
    const double stpc = *stp + r * (b->stx - *stp);
    const double stpq = *stp + dp / (dp - b->dx) * (b->stx - *stp);
    stpf = std::fabs(stpc - *stp) > std::fabs(stpq - *stp) ? stpc : stpq;
    *brackt = true;
  } else if (std::fabs(dp) < std::fabs(b->dx)) {
    *info = 3; bound = true;
    const double theta = 3 * (b->fx - fp) / (*stp - b->stx) + b->dx + dp;
    double gamma = cubic_gamma(theta, b->dx, dp, true);
    if (*stp > b->stx) gamma = -gamma;
    const double pp = gamma - dp + theta, q = gamma + (b->dx - dp) + gamma, r = pp / q;
    double stpc;
    if (r < 0 && gamma != 0) stpc = *stp + r * (b->stx - *stp);
    else stpc = *stp > b->stx ? stmax : stmin;
    const double stpq = *stp + dp / (dp - b->dx) * (b->stx - *stp);
    if (*brackt) stpf = std::fabs(*stp - stpc) < std::fabs(*stp - stpq) ? stpc : stpq;
    else stpf = std::fabs(*stp - stpc) > std::fabs(*stp - stpq) ? stpc : stpq;
  } else {
    *info = 4; bound = false;
    if (*brackt) {
      const double theta = 3 * (fp - b->fy) / (b->sty - *stp) + b->dy + dp;
      double gamma = cubic_gamma(theta, b->dy, dp, false);
      if (*stp > b->sty) gamma = -gamma;
      const double pp = gamma - dp + theta, q = gamma - dp + gamma + b->dy, r = pp / q;
      stpf = *stp + r * (b->sty - *stp);
    } else {
      stpf = *stp > b->stx ? stmax : stmin;
    }
  }
  if (fp > b->fx) {
    b->sty = *stp; b->fy = fp; b->dy = dp;
  } else {
    if (sgnd < 0.0) { b->sty = b->stx; b->fy = b->fx; b->dy = b->dx; }
    b->stx = *stp; b->fx = fp; b->dx = dp;
  }
  stpf = dmin(stmax, stpf);
  stpf = dmax(stmin, stpf);
  *stp = stpf;
  if (*brackt && bound) {
    if (b->sty > b->stx) *stp = dmin(b->stx + 0.66 * (b->sty - b->stx), *stp);
    else *stp = dmax(b->stx + 0.66 * (b->sty - b->stx), *stp);
  }
}

// mcsrch with the device evaluation inlined (constants alglibinternal.cpp:156-160;
// trimfunction after each evaluation as mincgiteration does, optimization.cpp:17594).
template <typename T>
static int line_search(DeviceCG<T>& cg, double* f, double dginit, double* stp, double gtol,
                       int* info, int* nfev, double trim, std::vector<double>* trace, double stp_in_x = 0.0,
                       bool pre_launched = false, double* dg_last = nullptr) {
  const double ftol = 0.001, xtol = 100 * 5E-16, stpmin = 1.0e-50, stpmax = 1.0e+50, p5 = 0.5,
               p66 = 0.66, xtrapf = 4.0;
  const int maxfev = 20;
  if (*stp < stpmin) *stp = stpmin;
  if (*stp > stpmax) *stp = stpmax;
  int infoc = 1;
  *info = 0;
  *nfev = 0;
  // On entry the base point is cg.xk; cg.x is scratch for the trial points.  The paths that try nothing
  // still leave x = base, as mcsrch does.
  // pre_launched: the evaluation at xk + stp_in_x * d is already queued (behind the pass that wrote that point)
  if (*stp <= 0 || dginit >= 0) {  // (dginit >= 0: not a descent direction)
    if (pre_launched) cg.discard_speculative();
    return cg.copy(cg.x, cg.xk);
  }
  bool brackt = false, stage1 = true;
  const double finit = *f, dgtest = ftol * dginit;
  double width = stpmax - stpmin, width1 = width / p5;
  int rc = SRMAP_OK;
  Bracket b = {0, finit, dginit, 0, finit, dginit};
  double stmin = 0, stmax = 0;
  for (;;) {
    if (brackt) { stmin = dmin(b.stx, b.sty); stmax = dmax(b.stx, b.sty); }
    else { stmin = b.stx; stmax = *stp + xtrapf * (*stp - b.stx); }
    if (*stp > stpmax) *stp = stpmax;
    if (*stp < stpmin) *stp = stpmin;
    if ((brackt && (*stp <= stmin || *stp >= stmax)) || *nfev >= maxfev - 1 || infoc == 0 ||
        (brackt && stmax - stmin <= xtol * stmax))
      *stp = b.stx;
    // stp_in_x: cg.x already holds xk + stp_in_x * d (written by the scaling pass) or the evaluation that forms it is
    // already queued (pre_launched); any other step is formed here
    const bool first_in_x = *nfev == 0 && stp_in_x != 0.0 && *stp == stp_in_x;
    if (*nfev == 0 && pre_launched && !first_in_x) { cg.discard_speculative(); pre_launched = false; }
    // the trial point x = xk + stp * d: formed by the evaluation itself as it loads its window where that is possible
    // (one n-vector pass less per trial point; x holds the point afterwards all the same), else by its own pass
    const bool fold_here = !first_in_x && cg.foldable;
    if (!first_in_x && !fold_here)
    {
      if ((cg.n & 3) == 0)
        hipLaunchKernelGGL(k_axpy_out4<T>, dim3((unsigned)((cg.n / 4 + 255) / 256)), dim3(256), 0, cg.st, cg.x, (const T*)cg.xk,
                           (const T*)cg.d, (T)*stp, cg.n / 4);
      else
        hipLaunchKernelGGL(k_axpy_out<T>, dim3(cg.blocks()), dim3(256), 0, cg.st, cg.x, (const T*)cg.xk,
                           (const T*)cg.d, (T)*stp, cg.n);
    }
    if (!(first_in_x && pre_launched)) {
      rc = fold_here ? cg.evaluate(cg.dk, nullptr, cg.xk, *stp) : cg.evaluate(cg.d);
      if (rc) return rc;
    }
    double h[2];
    rc = cg.fetch_f_gd(h);  // h[0] = g.d, h[1] = f
    if (rc) return rc;
    double dg = h[0];
    *f = h[1];
    if (trace) trace->push_back(*f);
    if (*f >= trim) {  // trimfunction: F = threshold, G = 0
      *f = trim;
      hipLaunchKernelGGL(k_fill<T>, dim3(cg.blocks()), dim3(256), 0, cg.st, cg.g, T(0), cg.n);
      dg = 0;
    }
    *info = 0;
    *nfev += 1;
    if (dg_last) *dg_last = dg;
    const double ftest1 = finit + *stp * dgtest;
    if ((brackt && (*stp <= stmin || *stp >= stmax)) || infoc == 0) *info = 6;
    if (*stp == stpmax && *f < finit && *f <= ftest1 && dg <= dgtest) *info = 5;
    if (*stp == stpmin && (*f >= finit || *f > ftest1 || dg >= dgtest)) *info = 4;
    if (*nfev >= maxfev) *info = 3;
    if (brackt && stmax - stmin <= xtol * stmax) *info = 2;
    if (*f < finit && *f <= ftest1 && std::fabs(dg) <= -gtol * dginit) *info = 1;
    if (*info != 0) {
      if (*info == 1 || *info == 5) {
        // ALGLIB additionally demotes to 6 when the point did not move
        // (sum (wa-x)^2 == 0); with stp > 0 and a unit d this cannot be 0
        // unless stp*d underflows against x, which we test through stp.
        if (*f >= finit || *stp == 0.0) *info = 6;
      }
      return SRMAP_OK;
    }
    if (stage1 && *f <= ftest1 && dg >= dmin(ftol, gtol) * dginit) stage1 = false;
    if (stage1 && *f <= b.fx && *f > ftest1) {
      const double fm = *f - *stp * dgtest;
      Bracket m = {b.stx, b.fx - b.stx * dgtest, b.dx - dgtest, b.sty, b.fy - b.sty * dgtest, b.dy - dgtest};
      mt_step(&m, stp, fm, dg - dgtest, &brackt, stmin, stmax, &infoc);
      b.stx = m.stx; b.sty = m.sty;
      b.fx = m.fx + m.stx * dgtest; b.fy = m.fy + m.sty * dgtest;
      b.dx = m.dx + dgtest; b.dy = m.dy + dgtest;
    } else {
      mt_step(&b, stp, *f, dg, &brackt, stmin, stmax, &infoc);
    }
    if (brackt) {
      if (std::fabs(b.sty - b.stx) >= p66 * width1) *stp = b.stx + p5 * (b.sty - b.stx);
      width1 = width;
      width = std::fabs(b.sty - b.stx);
    }
  }
}

struct CgResult { int type = 0, its = 0, nfev = 0; double f = 0; };

// mincgiteration (optimization.cpp:17137-17880), default configuration: no
// preconditioner, unit scales, cgtype = 1, no stpmax.  On return cg.x holds
// the accepted point XN.  trace (optional): f of every evaluation, in order.
template <typename T>
static int run_cg(DeviceCG<T>& cg, double epsg, double epsf, double epsx, int maxits, CgResult* out,
                  std::vector<double>* trace) {
  const double gtol = 0.3;
  const int rscountdownlen = 10;
  if (epsg == 0 && epsf == 0 && epsx == 0 && maxits == 0) epsx = 1.0E-6;
  const size_t n = cg.n;
  CgResult res;
  double f = 0, gg = 0;
  // the start point becomes the base point xk by exchanging the two buffers (no copy); x is trial scratch from here on:
  // every path below writes it before reading it (the trial points) or copies xk back into it (the early exits)
  std::swap(cg.xk, cg.x);
  {
    const int mode = (cg.comm && cg.shard && comm_world(cg.comm) > 1) ? cg.shard->mode : SRMAP_SHARD_NONE;
    cg.foldable = mode == SRMAP_SHARD_NONE && cg.fold_enabled && ztile_can_fold(cg.p);
  }
  int rc = cg.evaluate(nullptr, cg.xk);
  if (rc) return rc;
  // dk = -g (written as dn, swapped below), norms of dk; g.g = dk.dk comes with them
  rc = cg.direction(nullptr, 0.0, true);
  if (rc) return rc;
  if (cg.fused()) {  // the direction pass published {f -> hs[0]; max|dk|, dk.dk, g.dk -> hs[8..10]} itself
    rc = cg.wait_tag();
    if (rc) return rc;
    f = cg.hs[0];
  } else {
    // fetch f and the direction's sums (already reduced on the device) with one wait
    double h[1];
    rc = cg.finish(0, false, true, h, 3);
    if (rc) return rc;
    cg.dir_tag = cg.tag;
    f = h[0];
  }
  gg = cg.hs[9];  // g.g = dk.dk
  if (trace) trace->push_back(f);
  std::swap(cg.dk, cg.dn);
  const double trim = 10 * (std::fabs(f) + 1);
  if (std::sqrt(gg) <= epsg) { res.type = 4; res.f = f; *out = res; return cg.copy(cg.x, cg.xk); }
  res.nfev = 1;
  double fold = f, lastgoodstep = 1.0;
  int rstimer = rscountdownlen;
  for (;;) {
    // d = dk s1 s2 (linminnormalized); g.d and d.d follow from the sums of the pass that produced dk.  x = xk is not
    // materialised: every trial point x = xk + stp * d is written by the line search -- by the evaluation itself, from dk
    // and the norms on the device, where the tile kernel can (foldable), else from the d a scaling pass stores.
    double stp = 1.0, dginit = 0, dd = 0;
    double gdk = 0, ns1 = 1, ns2 = 1;  // g.dk at xk and the two scale factors, for the beta denominator below
    bool pre_launched = false, g_swapped = false;
    // the first step is lastgoodstep unless that is 0 (then it comes from the direction's norms)
    const double stp_pre = (lastgoodstep != 0 && lastgoodstep >= 1.0e-50 && lastgoodstep <= 1.0e+50) ? lastgoodstep : 0.0;
    double stp_ready = 0.0;  // the step whose trial point is already in cg.x and / or whose evaluation is already queued
    {
      if (!cg.foldable) {
        rc = cg.normalize(stp_pre);
        if (rc) return rc;
        stp_ready = stp_pre;
      }
      // The line search's first trial evaluation is queued NOW, behind the passes above, and runs while the host waits
      // for the direction's sums and decides (mcsrch tries stp first whenever g.d < 0; otherwise line_search discards the
      // evaluation): no host round trip in front of it.
      if (stp_pre != 0.0 && cg.chained()) {
        std::swap(cg.g, cg.gp);  // gp = gradient at xk
        g_swapped = true;
        rc = cg.foldable ? cg.evaluate(cg.dk, nullptr, cg.xk, stp_pre) : cg.evaluate(cg.d);
        if (rc) return rc;
        pre_launched = true;
        stp_ready = stp_pre;
      }
      double mx = 0, ss = 0, gdn = 0;
      rc = cg.wait_dir(&mx, &ss, &gdn);
      if (rc) return rc;
      double s1, s2;
      norm_factors(mx, ss, s1, s2);
      dginit = (gdn * s1) * s2;
      dd = ((ss * s1) * s1) * (s2 * s2);
      gdk = gdn; ns1 = s1; ns2 = s2;
      if (mx != 0) { stp = stp / s1; stp = stp / s2; }
    }
    if (lastgoodstep != 0) stp = lastgoodstep;
    int mcinfo = 0, nfev = 0;
    if (!g_swapped) std::swap(cg.g, cg.gp);  // gp = gradient at xk; the trial evaluations write g
    double dg_acc = 0;  // g.d at the last trial point
    rc = line_search(cg, &f, dginit, &stp, gtol, &mcinfo, &nfev, trim, trace, stp_ready, pre_launched, &dg_acc);
    if (rc) return rc;
    if (nfev == 0) std::swap(cg.g, cg.gp);  // nothing was evaluated: g stays the gradient at xk, as in mcsrch
    double betak = 0;
    // One-launch scheme: the direction pass is queued right behind the pass that reduces the beta sums -- beta itself is
    // formed by that pass's finishing thread (k_beta_dots) -- and runs while the host waits for the sums it needs for
    // the stopping rules.  (mincg's periodic restart is known beforehand; `direction` was always launched before the
    // rules are looked at.)
    const bool chain = cg.chained();
    const int restart = (res.its > 0 && res.its % (3 + (long long)n) == 0) ? 1 : 0;
    if (mcinfo == 1) {
      // yk = g - gp ; vv = yk.dk ; betady = g.g/vv ; betahs = g.yk/vv.  vv = g.dk - gp.dk from sums already on the
      // host: gp.dk is the direction pass's g.dn, g.dk = (g.d) / (s2 s1) with the accepted evaluation's g.d (k_beta_dots)
      const double vv = (dg_acc / ns2) / ns1 - gdk;
      double* beta_dst = chain ? cg.dscal + 8 : (double*)nullptr;
      // host-paced passes: the pass also sums y.dk directly (one more vector read) and the deviation of the derived
      // denominator from it is recorded (srmap_problem_selfcheck; tests/test_gpu_solve_parity.py)
      const T* dk_chk = chain ? (const T*)nullptr : (const T*)cg.dk;
      if (cg.vec())
        hipLaunchKernelGGL((k_beta_dots<T, DeviceCG<T>::kVec>), dim3(cg.nb()), dim3(256), 0, cg.st, (const T*)cg.gp, (const T*)cg.g,
                           n, cg.ow, cg.part, cg.fin_host(false), beta_dst, restart, vv, dk_chk);
      else
        hipLaunchKernelGGL((k_beta_dots<T, 1>), dim3(cg.nb()), dim3(256), 0, cg.st, (const T*)cg.gp, (const T*)cg.g,
                           n, cg.ow, cg.part, cg.fin_host(false), beta_dst, restart, vv, dk_chk);
      if (chain) {
        const double tag_beta = cg.tag;
        rc = cg.direction(cg.dk, 0.0, false, cg.dscal + 8);  // dn, sums of dn (device)
        if (rc) return rc;
        rc = cg.wait_tag(tag_beta);
        if (rc) return rc;
        gg = cg.hs[0];
      } else {
        double h[3];
        rc = cg.finish(3, false, false, h);
        if (rc) return rc;
        betak = dmax(0.0, dmin(h[0] / vv, h[1] / vv));
        gg = h[0];
        if (h[2] != 0.0 && std::isfinite(h[2]) && std::isfinite(vv))
          cg.p->selfcheck_beta_den = dmax(cg.p->selfcheck_beta_den, std::fabs(vv - h[2]) / std::fabs(h[2]));
      }
    } else {
      if (cg.vec())
        hipLaunchKernelGGL((k_dot<T, DeviceCG<T>::kVec>), dim3(cg.nb()), dim3(256), 0, cg.st, (const T*)cg.g, (const T*)cg.g, n, cg.ow,
                           cg.part, cg.fin_host(false));
      else
        hipLaunchKernelGGL((k_dot<T, 1>), dim3(cg.nb()), dim3(256), 0, cg.st, (const T*)cg.g, (const T*)cg.g, n, cg.ow, cg.part,
                           cg.fin_host(false));
      if (chain) {
        const double tag_gg = cg.tag;
        rc = cg.direction(cg.dk, 0.0);  // beta = 0
        if (rc) return rc;
        rc = cg.wait_tag(tag_gg);
        if (rc) return rc;
        gg = cg.hs[0];
      } else {
        double h[1];
        rc = cg.finish(1, false, false, h);
        if (rc) return rc;
        gg = h[0];
      }
    }
    if (restart) betak = 0;
    if (mcinfo == 1 || mcinfo == 5) rstimer = rscountdownlen; else rstimer -= 1;
    if (!chain) {
      rc = cg.direction(cg.dk, betak);  // dn, norms of dn (device)
      if (rc) return rc;
    }
    const double lastscaledstep = stp * std::sqrt(dd);
    if (mcinfo == 1) lastgoodstep = stp * std::sqrt(dd);
    if (!std::isfinite(gg) || !std::isfinite(f)) { res.type = -8; break; }
    res.nfev += nfev;
    res.its += 1;
    if (res.its >= maxits && maxits > 0) { res.type = 5; break; }
    if (std::sqrt(gg) <= epsg) { res.type = 4; break; }
    if (fold - f <= epsf * dmax(std::fabs(fold), dmax(std::fabs(f), 1.0))) { res.type = 1; break; }
    if (lastscaledstep <= epsx) { res.type = 2; break; }
    if (rstimer <= 0) { res.type = 7; break; }
    std::swap(cg.xk, cg.x);    // xk <- accepted point; the old xk becomes trial scratch
    std::swap(cg.dk, cg.dn);   // dk <- new direction
    fold = f;
  }
  res.f = f;
  *out = res;
  return SRMAP_OK;
}

template <typename T>
static int solve_typed(srmap_problem* p, srmap_comm* comm, const srmap_shard_desc* shard, const srmap_irls_options* opt,
                       const double* x0, double* x_out, srmap_solve_report* report) {
  if (!p->have_obs) return set_error(p->ctx, SRMAP_EINVAL, "cannot super-resolve with 0 low-res images");
  const Geometry& geo = p->geo;
  const size_t N = (size_t)geo.W * geo.H;
  const int C = geo.C;
  const int mode = (comm && shard && comm_world(comm) > 1) ? shard->mode : SRMAP_SHARD_NONE;
  if (mode != SRMAP_SHARD_NONE && opt->split_channels)
    return set_error(p->ctx, SRMAP_EUNSUPPORTED, "split_channels solves are independent per channel: run them unsharded");
  if (mode == SRMAP_SHARD_ROWS &&
      (shard->own_row0 < 0 || shard->own_row1 > geo.H || shard->own_row0 >= shard->own_row1 ||
       shard->own_row0 != geo.cr0 || shard->own_row1 != geo.cr1))
    return set_error(p->ctx, SRMAP_EINVAL, "row shard: owned rows must equal the problem's cost rows");
  if ((mode == SRMAP_SHARD_CHANNELS || mode == SRMAP_SHARD_GRID) &&
      (shard->own_ch0 < 0 || shard->own_ch1 > C || shard->own_ch0 >= shard->own_ch1))
    return set_error(p->ctx, SRMAP_EINVAL, "channel shard: bad owned channel range");
  if (mode == SRMAP_SHARD_GRID && (shard->frame_groups < 1 || comm_world(comm) % shard->frame_groups != 0 ||
                                   (shard->frame_groups > 1 && !shard->frame_comm)))
    return set_error(p->ctx, SRMAP_EINVAL, "grid shard: frame_groups must divide the world size and frame_comm must be given");
  // GRID: the unknowns of a channel block are replicated over its frame groups; group 0 counts them in the reductions
  const bool grid_replica = mode == SRMAP_SHARD_GRID && shard->frame_groups > 1 && (comm_rank(comm) % shard->frame_groups) != 0;
  const int per_split = opt->split_channels ? 1 : C;
  const int rounds = C / per_split;
  const size_t npts = (size_t)per_split * N;
  srmap_irls_options o = *opt;
  double lambda_sum = 0.0;
  for (int r = 0; r < p->nreg; ++r) lambda_sum += p->reg[r].lambda;
  {  // AdjustThresholdsAdaptively (map_solver.cpp:16-26, irls_map_solver.cpp:161-171) on the JOINT problem size
    double params = (double)(int)npts;
    if (mode == SRMAP_SHARD_ROWS || mode == SRMAP_SHARD_CHANNELS || mode == SRMAP_SHARD_GRID) {
      double own = mode == SRMAP_SHARD_ROWS ? (double)C * geo.W * (shard->own_row1 - shard->own_row0)
                                            : (grid_replica ? 0.0 : (double)(shard->own_ch1 - shard->own_ch0) * N);
      // every rank must derive the same thresholds: total parameter count = sum of the owned counts
      double* tmp = nullptr;
      SRMAP_HIP(p->ctx, hipMalloc((void**)&tmp, sizeof(double)));
      SRMAP_HIP(p->ctx, hipMemcpy(tmp, &own, sizeof(double), hipMemcpyHostToDevice));
      int rc0 = comm_allreduce(comm, tmp, 1, SRMAP_F64, 0, p->ctx->stream);
      if (rc0 == SRMAP_OK) {
        SRMAP_HIP(p->ctx, hipStreamSynchronize(p->ctx->stream));
        SRMAP_HIP(p->ctx, hipMemcpy(&own, tmp, sizeof(double), hipMemcpyDeviceToHost));
      }
      (void)hipFree(tmp);
      if (rc0) return rc0;
      params = (double)(int)own;
    }
    const double scale = params * lambda_sum;
    if (!(scale < 1.0)) {
      o.gradient_norm_threshold *= scale;
      o.cost_decrease_threshold *= scale;
      o.parameter_variation_threshold *= scale;
      o.irls_cost_difference_threshold *= scale;
    }
  }
  srmap_solve_report rep = {0, 0, 0, 0, 0.0, 0.0, 0.0, 0};
  hipStream_t st = p->ctx->stream;
  DeviceCG<T> cg;
  cg.p = p; cg.st = st; cg.n = npts; cg.comm = comm; cg.shard = shard;
  cg.chain_enabled = o.host_paced_passes == 0;
  cg.fold_enabled = o.host_paced_passes == 0;
  cg.ow.on = 0; cg.ow.e0 = 0; cg.ow.e1 = npts; cg.ow.W = geo.W; cg.ow.H = geo.H; cg.ow.r0 = 0; cg.ow.r1 = geo.H;
  if (mode == SRMAP_SHARD_ROWS) { cg.ow.on = 1; cg.ow.r0 = shard->own_row0; cg.ow.r1 = shard->own_row1; cg.reduce_scalars = true; }
  if (mode == SRMAP_SHARD_CHANNELS || mode == SRMAP_SHARD_GRID) {
    cg.ow.on = 1; cg.ow.e0 = (size_t)shard->own_ch0 * N; cg.ow.e1 = grid_replica ? cg.ow.e0 : (size_t)shard->own_ch1 * N;
    cg.reduce_scalars = true;
  }
  int rc = cg.alloc();
  // IRLS weights live in the problem's RegSpec (full [C][H][W]); make sure they exist.
  for (int r = 0; r < p->nreg && rc == SRMAP_OK; ++r) {
    if (!p->reg[r].weights) {
      hipError_t e = hipMalloc(&p->reg[r].weights, p->hr_count() * sizeof(T));
      if (e != hipSuccess) rc = set_error(p->ctx, SRMAP_ENOMEM, "hipMalloc failed");
    }
  }
  const int saved_c0 = p->view_c0, saved_C = p->view_C;
  for (int round = 0; round < rounds && rc == SRMAP_OK; ++round) {
    const int c0 = round * per_split;
    if (opt->split_channels) { p->view_c0 = c0; p->view_C = per_split; }
    Geometry vg = geo;
    vg.C = per_split;
    rc = convert_upload(p, x0 + (size_t)c0 * N, cg.x, npts, st);
    if (rc) break;
    // w <- 1  (irls_map_solver.cpp:66-74)
    for (int r = 0; r < p->nreg; ++r)
      hipLaunchKernelGGL(k_fill<T>, dim3(cg.blocks()), dim3(256), 0, st, (T*)p->reg[r].weights + (size_t)c0 * N, T(1), npts);
    double previous_cost = INFINITY;
    double cost_difference = o.irls_cost_difference_threshold + 1.0;
    int ran = 0;
    SRMAP_HIP(p->ctx, hipStreamSynchronize(st));
    const auto t_loop0 = std::chrono::steady_clock::now();
    while (std::fabs(cost_difference) >= o.irls_cost_difference_threshold) {
      CgResult cr;
      rc = run_cg(cg, o.gradient_norm_threshold, o.cost_decrease_threshold, o.parameter_variation_threshold,
                  o.max_num_solver_iterations, &cr, nullptr);
      if (rc) break;
      rep.cg_iterations += cr.its;
      rep.last_termination = cr.type;
      rep.final_cost = cr.f;
      if (p->nreg == 0) { ran++; break; }
      // w = 1/max(1e-5, reg(x)), :128-143 -- on fresh halos (the weights of a halo plane / halo rows feed the
      // owned gradient through the neighbour terms)
      rc = shard_exchange_x(p, comm, mode == SRMAP_SHARD_NONE ? nullptr : shard, cg.x, st);
      if (rc) break;
      for (int r = 0; r < p->nreg; ++r) {
        rc = launch_reg_weights<T>(p, vg, p->reg[r], (const T*)cg.x, (T*)p->reg[r].weights + (size_t)c0 * N, st);
        if (rc) break;
      }
      if (rc) break;
      cost_difference = previous_cost - cr.f;
      previous_cost = cr.f;
      ran++;
      if (o.max_num_irls_iterations > 0 && ran >= o.max_num_irls_iterations) break;
    }
    if (rc) break;
    (void)hipStreamSynchronize(st);
    rep.loop_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop0).count();
    rep.irls_rounds += ran;
    rc = shard_exchange_x(p, comm, mode == SRMAP_SHARD_NONE ? nullptr : shard, cg.x, st);  // x_out carries valid halos
    if (rc) break;
    rc = convert_download(p, cg.x, x_out + (size_t)c0 * N, npts, st);
  }
  rep.evaluations = cg.evaluations;
  rep.wait_seconds = cg.wait_seconds;
  rep.waits = cg.waits;
  p->view_c0 = saved_c0;
  p->view_C = saved_C;
  {
    // A reduction that gave up waiting for a workgroup (the tile kernel's in-kernel finish: sticky word d_cost[6]; a CG
    // pass: dscal[15]) left NaN sums behind -- the stopping rules ended the run -- and its granules un-re-armed.  The
    // evaluations inside a solve never look at the word (they pass no cost pointer): look now, re-initialise, report.
    // Both kinds of finisher also raise the host-mapped word hs[13] (wait_tag ends the solve on it at once): no device
    // copy on the successful path.
    (void)hipStreamSynchronize(st);
    if (cg.hs != nullptr && cg.hs[13] != 0.0) {
      cg.hs[13] = 0.0;
      (void)hipDeviceSynchronize();
      ztile_rearm(p);
      if (p->d_cost) (void)hipMemset(p->d_cost + 6, 0, sizeof(double));
      if (rc == SRMAP_OK) rc = set_error(p->ctx, SRMAP_EHIP, "a device-side reduction timed out waiting for a workgroup during the solve (device fault or a wedged queue)");
    }
  }
  cg.release();
  if (report) *report = rep;
  return rc;
}

int solve_impl(srmap_problem* p, srmap_comm* comm, const srmap_shard_desc* shard, const srmap_irls_options* o,
               const double* x0, double* x_out, srmap_solve_report* rep) {
  if (p->dtype == SRMAP_F32) return solve_typed<float>(p, comm, shard, o, x0, x_out, rep);
  return solve_typed<double>(p, comm, shard, o, x0, x_out, rep);
}

// One nonlinear-CG run (no IRLS re-weighting) with the f of every evaluation recorded: the trajectory the tests
// compare with ALGLIB's mincg on the same objective (tests/test_gpu_parity.py).
template <typename T>
static int cg_trace_typed(srmap_problem* p, double epsg, double epsf, double epsx, int maxits, const double* x0,
                          double* x_out, int* iterations, int* nfev, int* termination, double* f_trace, int trace_cap,
                          int* trace_len) {
  const size_t npts = p->hr_count();
  DeviceCG<T> cg;
  cg.p = p; cg.st = p->ctx->stream; cg.n = npts;
  cg.ow.on = 0; cg.ow.e0 = 0; cg.ow.e1 = npts; cg.ow.W = p->geo.W; cg.ow.H = p->geo.H; cg.ow.r0 = 0; cg.ow.r1 = p->geo.H;
  int rc = cg.alloc();
  if (rc == SRMAP_OK) rc = convert_upload(p, x0, cg.x, npts, cg.st);
  CgResult cr;
  std::vector<double> tr;
  if (rc == SRMAP_OK) rc = run_cg(cg, epsg, epsf, epsx, maxits, &cr, &tr);
  if (rc == SRMAP_OK) rc = convert_download(p, cg.x, x_out, npts, cg.st);
  cg.release();
  if (rc) return rc;
  if (iterations) *iterations = cr.its;
  if (nfev) *nfev = cr.nfev;
  if (termination) *termination = cr.type;
  const int m = (int)tr.size() < trace_cap ? (int)tr.size() : trace_cap;
  for (int i = 0; i < m; ++i) f_trace[i] = tr[i];
  if (trace_len) *trace_len = (int)tr.size();
  return SRMAP_OK;
}

}  // namespace srmap

using namespace srmap;

extern "C" {

int srmap_eval_sharded_device(srmap_problem* p, srmap_comm* comm, const srmap_shard_desc* shard, unsigned terms,
                              void* x_dev, void* g_dev, double* cost, void* hip_stream) {
  if (!p || !x_dev) return SRMAP_EINVAL;
  SRMAP_HIP(p->ctx, hipSetDevice(p->ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : p->ctx->stream;
  int rc = shard_eval(p, comm, shard, terms, x_dev, g_dev, st);
  if (rc) return rc;
  if (cost) {
    const int mode = (comm && shard && comm_world(comm) > 1) ? shard->mode : SRMAP_SHARD_NONE;
    if (mode == SRMAP_SHARD_ROWS || mode == SRMAP_SHARD_CHANNELS || mode == SRMAP_SHARD_GRID) {
      rc = comm_allreduce(comm, p->d_cost, 1, SRMAP_F64, 0, st);
      if (rc) return rc;
    }
    SRMAP_HIP(p->ctx, hipMemcpyAsync(cost, p->d_cost, sizeof(double), hipMemcpyDeviceToHost, st));
    SRMAP_HIP(p->ctx, hipStreamSynchronize(st));
  }
  return SRMAP_OK;
}

int srmap_cg_trace(srmap_problem* p, double epsg, double epsf, double epsx, int maxits, const double* x0, double* x_out,
                   int* iterations, int* nfev, int* termination, double* f_trace, int trace_cap, int* trace_len) {
  if (!p || !x0 || !x_out || (trace_cap > 0 && !f_trace)) return SRMAP_EINVAL;
  SRMAP_HIP(p->ctx, hipSetDevice(p->ctx->device));
  if (!p->have_obs) return set_error(p->ctx, SRMAP_EINVAL, "no observations set");
  if (p->dtype == SRMAP_F32)
    return cg_trace_typed<float>(p, epsg, epsf, epsx, maxits, x0, x_out, iterations, nfev, termination, f_trace, trace_cap, trace_len);
  return cg_trace_typed<double>(p, epsg, epsf, epsx, maxits, x0, x_out, iterations, nfev, termination, f_trace, trace_cap, trace_len);
}

}  // extern "C"
