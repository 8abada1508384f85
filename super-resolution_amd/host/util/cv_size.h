// cv::Size as the reference's MAP-path signatures use it (Regularizer,
// MapSolver::GetImageSize).  With OpenCV on the include path the real type is
// used; otherwise the facade carries this two-int equivalent (OpenCV is not
// needed by the HIP library itself).
#pragma once
#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#define SRMAP_HAVE_OPENCV 1
#endif
#endif
#ifndef SRMAP_HAVE_OPENCV
namespace cv {
struct Size {
  int width = 0, height = 0;
  Size() = default;
  Size(int w, int h) : width(w), height(h) {}
  int area() const { return width * height; }
  bool operator==(const Size& o) const { return width == o.width && height == o.height; }
  bool operator!=(const Size& o) const { return !(*this == o); }
};
}  // namespace cv
#endif
