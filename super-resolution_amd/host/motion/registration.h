// registration::TranslationalRegistration (src/motion/registration.h:19-22, registration.cpp:161-201): shifts of a
// list of images relative to the first one, as a MotionShiftSequence.  Same signature and calling conventions as the
// reference -- NOT the same estimator: a dense pure-translation search (up to a quarter of the frame) with no
// outlier rejection, where the reference fits a feature-based RANSAC homography / rigid transform and keeps its
// translation.  Rotation, scale or periodic texture between real frames can give different (wrong) shifts without
// an error; include/srmap.h (srmap_register_translational_ex) reports a per-frame quality to check, and
// TranslationalRegistrationWithQuality below exposes it.  (channel 0 is the registration image, registration.cpp:41-46; an empty list gives an empty sequence,
// :165-168; image 0 gets (0, 0), :170-172; failure to determine a shift is a CHECK failure, :193-194).  The
// estimate itself comes from the GPU (srmap_register_translational, csrc/registration.hip) instead of the
// reference's OpenCV feature pipeline; the contract is the reference's own test (test/test_registration.cpp:
// shifts applied with MotionModule recovered to 0.01 px).
#pragma once
#include <vector>

#include "image/image_data.h"
#include "motion/motion_shift.h"
#include "util/srmap_host.h"

namespace super_resolution {
namespace registration {

inline MotionShiftSequence TranslationalRegistration(const std::vector<ImageData>& images) {
  if (images.empty()) {
    std::fprintf(stderr, "WARNING: No images given. Returning an empty motion sequence.\n");
    return MotionShiftSequence();
  }
  const cv::Size size = images[0].GetImageSize();
  const size_t npx = static_cast<size_t>(size.width) * size.height;
  std::vector<double> stack(npx * images.size());
  for (size_t i = 0; i < images.size(); ++i) {
    if (images[i].GetNumChannels() < 1 || images[i].GetImageSize().width != size.width ||
        images[i].GetImageSize().height != size.height)
      srmap_host::Fail("registration needs images of one size with at least one channel");
    const double* ch = images[i].GetChannelData(0);
    std::copy(ch, ch + npx, stack.begin() + i * npx);
  }
  std::vector<double> xy(2 * images.size());
  srmap_host::Check(srmap_register_translational(srmap_host::Context(), static_cast<int>(images.size()), size.width,
                                                 size.height, stack.data(), xy.data()),
                    "Could not determine motion shift between images.");
  std::vector<MotionShift> shifts;
  for (size_t i = 0; i < images.size(); ++i) shifts.push_back(MotionShift(xy[2 * i], xy[2 * i + 1]));
  return MotionShiftSequence(shifts);
}

// The same with the estimator's per-image quality: quality[2i] = separation of the coarse minimum (near 1 = clear,
// near 0 = ambiguous), quality[2i + 1] = RMS residual at the returned shift.
inline MotionShiftSequence TranslationalRegistrationWithQuality(const std::vector<ImageData>& images,
                                                                std::vector<double>* quality) {
  if (images.empty()) return MotionShiftSequence();
  const cv::Size size = images[0].GetImageSize();
  const size_t npx = static_cast<size_t>(size.width) * size.height;
  std::vector<double> stack(npx * images.size());
  for (size_t i = 0; i < images.size(); ++i) {
    if (images[i].GetNumChannels() < 1 || images[i].GetImageSize().width != size.width ||
        images[i].GetImageSize().height != size.height)
      srmap_host::Fail("registration needs images of one size with at least one channel");
    const double* ch = images[i].GetChannelData(0);
    std::copy(ch, ch + npx, stack.begin() + i * npx);
  }
  std::vector<double> xy(2 * images.size());
  quality->assign(2 * images.size(), 0.0);
  srmap_host::Check(srmap_register_translational_ex(srmap_host::Context(), static_cast<int>(images.size()), size.width,
                                                    size.height, stack.data(), xy.data(), quality->data()),
                    "Could not determine motion shift between images.");
  std::vector<MotionShift> shifts;
  for (size_t i = 0; i < images.size(); ++i) shifts.push_back(MotionShift(xy[2 * i], xy[2 * i + 1]));
  return MotionShiftSequence(shifts);
}

}  // namespace registration
}  // namespace super_resolution
