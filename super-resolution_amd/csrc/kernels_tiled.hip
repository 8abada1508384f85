// kernels_tiled.hip -- LDS-tiled fused evaluation (placeholder until the
// fused kernel lands; AUTO falls back to the direct kernels).
#include "srmap_internal.hpp"

namespace srmap {

bool tiled_plan(srmap_problem*) { return false; }

template <typename T>
int launch_eval_tiled(srmap_problem* p, const Geometry&, int, unsigned, const T*, T*, double*, int*,
                      hipStream_t) {
  return set_error(p->ctx, SRMAP_EUNSUPPORTED, "tiled kernels not built");
}
template int launch_eval_tiled<float>(srmap_problem*, const Geometry&, int, unsigned, const float*, float*,
                                      double*, int*, hipStream_t);
template int launch_eval_tiled<double>(srmap_problem*, const Geometry&, int, unsigned, const double*,
                                       double*, double*, int*, hipStream_t);

}  // namespace srmap
