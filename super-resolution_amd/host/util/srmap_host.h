// Shared plumbing of the C++ facade: one srmap context per process and the
// reference's error convention (a failed CHECK aborts the process with a
// message -- glog semantics; the C ABI itself never aborts).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "srmap.h"

namespace super_resolution {
namespace srmap_host {

inline srmap_ctx* Context() {
  static srmap_ctx* ctx = [] {
    srmap_ctx* c = nullptr;
    const char* dev = std::getenv("SRMAP_DEVICE");
    if (srmap_ctx_create(dev ? std::atoi(dev) : 0, &c) != SRMAP_OK) {
      std::fprintf(stderr, "Check failed: no usable MI355X / HIP device (this library has no CPU path)\n");
      std::abort();
    }
    return c;
  }();
  return ctx;
}

// CHECK-style failure of a HOST-side precondition (bad size, null image, channel mismatch): print the given
// message and abort.  Does not touch the GPU context, so argument errors read the same on a box without a device.
[[noreturn]] inline void Fail(const char* what) {
  std::fprintf(stderr, "Check failed: %s\n", what);
  std::abort();
}

// CHECK-style on the status of a LIBRARY call: abort with the library's message.
inline void Check(int status, const char* what) {
  if (status == SRMAP_OK) return;
  std::fprintf(stderr, "Check failed: %s: %s (srmap status %d)\n", what, srmap_last_error(Context()), status);
  std::abort();
}

struct ProblemDeleter {
  void operator()(srmap_problem* p) const { srmap_problem_destroy(p); }
};
using ProblemPtr = std::unique_ptr<srmap_problem, ProblemDeleter>;

// Parameters of one operator chain  D(scale) . B(ksize, sigma) . M(shifts).
struct ChainParams {
  int scale = 1;
  std::vector<double> shifts_xy;  // empty = no MotionModule
  int frames = 1;
  int blur_ksize = 0;
  double blur_sigma = 0.0;
};

inline ProblemPtr MakeProblem(const ChainParams& c, int width, int height, int channels) {
  srmap_problem_desc d;
  d.hr_width = width; d.hr_height = height; d.channels = channels;
  d.frames = c.shifts_xy.empty() ? c.frames : static_cast<int>(c.shifts_xy.size() / 2);
  d.scale = c.scale;
  d.shifts_xy = c.shifts_xy.empty() ? nullptr : c.shifts_xy.data();
  d.blur_ksize = c.blur_ksize; d.blur_sigma = c.blur_sigma;
  d.dtype = SRMAP_F64;  // the reference computes in double
  srmap_problem* p = nullptr;
  Check(srmap_problem_create(Context(), &d, &p), "srmap_problem_create");
  return ProblemPtr(p);
}

}  // namespace srmap_host
}  // namespace super_resolution
