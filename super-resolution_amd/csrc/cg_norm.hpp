// cg_norm.hpp -- the normalised CG direction, element by element, wherever it is formed.
//
// linminnormalized (alglibinternal.cpp:12165-12196) scales the direction by s1 = 1 / max|dn|, then by
// s2 = 1 / sqrt(sum (dn s1)^2); the sum is taken as (dn.dn) * s1^2 from the pass that produced dn.  Every consumer of the
// normalised direction -- the pass that stores it (solver.hip k_normalize), the evaluation that forms a trial point from
// the unnormalised direction directly (kernels_ztile.hip, fold_norms) and the host's step scaling (run_cg) -- derives the
// two factors from the reduced norms with THESE IEEE operations and applies them with norm_elem: the same bits wherever
// d_i is formed.
#pragma once
#include <cmath>

#include <hip/hip_runtime.h>

namespace srmap {

__host__ __device__ __forceinline__ void norm_factors(double mx, double ss, double& s1, double& s2) {
  s1 = 1.0;
  s2 = 1.0;
  if (mx != 0.0) { s1 = 1.0 / mx; s2 = 1.0 / sqrt(ss * s1 * s1); }
}
template <typename T>
__host__ __device__ __forceinline__ T norm_elem(T v, double mx, double s1, double s2) {
  return mx != 0.0 ? (T)(((double)v * s1) * s2) : v;
}

#if defined(__HIPCC__)
// How an evaluation reads the search direction: the vector as given, or -- norms given -- the normalised direction formed
// from the unnormalised one (uniform values, kept in scalar registers).
struct DirScale {
  double mx, s1, s2;
  bool on;
};
__device__ __forceinline__ double uniform_d(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ DirScale dir_scale(const double* __restrict__ norms) {
  DirScale ds{0.0, 1.0, 1.0, false};
  if (norms != nullptr) {  // uniform
    double s1, s2;
    const double mx = norms[0];
    norm_factors(mx, norms[1], s1, s2);
    ds.mx = uniform_d(mx); ds.s1 = uniform_d(s1); ds.s2 = uniform_d(s2); ds.on = true;
  }
  return ds;
}
template <typename T>
__device__ __forceinline__ T dir_elem(T v, const DirScale& ds) { return ds.on ? norm_elem<T>(v, ds.mx, ds.s1, ds.s2) : v; }
#endif

}  // namespace srmap
