// ImageData -- the planar double image container of the reference
// (src/image/image_data.h:109-353), reduced to what the MAP gradient path
// touches: channels as contiguous H x W double planes, the pixel-array
// constructor (image_data.cpp:244-265), AddChannel(const double*, Size)
// (:298-308), GetChannelData / GetMutableChannelData (:526-537), deep copies.
// Colour-space conversion, visualisation and OpenCV cv::Mat interop are outside
// the path.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "util/cv_size.h"
#include "util/srmap_host.h"

namespace super_resolution {

// ImageData::ResizeImage methods (image_data.h:25-45); ADDITIVE/CUBIC are not on the host path here
enum ResizeInterpolationMethod { INTERPOLATE_NEAREST, INTERPOLATE_LINEAR };

class ImageData {
 public:
  ImageData() {}
  // pixel_values: planar [C][H][W], copied.
  ImageData(const double* pixel_values, const cv::Size& size, const int num_channels = 1)
      : image_size_(size) {
    if (!pixel_values || size.area() <= 0 || num_channels <= 0) srmap_host::Check(SRMAP_EINVAL, "ImageData");
    const size_t n = static_cast<size_t>(size.area());
    for (int c = 0; c < num_channels; ++c) channels_.emplace_back(pixel_values + c * n, pixel_values + (c + 1) * n);
  }
  void AddChannel(const double* pixel_values, const cv::Size& size) {
    if (!channels_.empty() && size != image_size_) srmap_host::Check(SRMAP_EINVAL, "AddChannel: size mismatch");
    image_size_ = size;
    channels_.emplace_back(pixel_values, pixel_values + size.area());
  }
  int GetNumChannels() const { return static_cast<int>(channels_.size()); }
  cv::Size GetImageSize() const { return image_size_; }
  int GetNumPixels() const { return image_size_.area(); }
  const double* GetChannelData(const int index) const { return channels_.at(index).data(); }
  double* GetMutableChannelData(const int index) { return channels_.at(index).data(); }
  double GetPixelValue(const int channel, const int index) const { return channels_.at(channel).at(index); }

  // ResizeImage(scale_factor, method) (image_data.cpp:310-364): new size = (int)(size * scale_factor);
  // cv::resize semantics restated for CV_64F: source coordinate of destination pixel d is
  // (d + 0.5) / fx - 0.5 with fx = new / old; NEAREST takes floor(d / fx) clamped; LINEAR blends the two
  // neighbours with the fraction rounded to float (OpenCV keeps the coefficients as float), clamping both
  // taps to the image (replicated border).  OpenCV is absent here: parity unpinned, used only for the
  // initial estimate and the "upsampled" baseline of the CLI.
  void ResizeImage(const double scale_factor, const ResizeInterpolationMethod method = INTERPOLATE_LINEAR) {
    if (channels_.empty() || !(scale_factor > 0)) srmap_host::Check(SRMAP_EINVAL, "ResizeImage");
    const int ow = image_size_.width, oh = image_size_.height;
    const int nw = static_cast<int>(ow * scale_factor), nh = static_cast<int>(oh * scale_factor);
    if (nw <= 0 || nh <= 0) srmap_host::Check(SRMAP_EINVAL, "ResizeImage: images must have a positive size");
    const double inv_fx = static_cast<double>(ow) / nw, inv_fy = static_cast<double>(oh) / nh;
    std::vector<int> x0(nw), y0(nh);
    std::vector<float> ax(nw), ay(nh);
    auto taps = [&](int n_new, int n_old, double inv_f, std::vector<int>& i0, std::vector<float>& a) {
      for (int d = 0; d < n_new; ++d) {
        if (method == INTERPOLATE_NEAREST) {
          i0[d] = std::min(static_cast<int>(std::floor(d * inv_f)), n_old - 1);
          a[d] = 0.f;
          continue;
        }
        float f = static_cast<float>((d + 0.5) * inv_f - 0.5);
        int i = static_cast<int>(std::floor(f));
        f -= static_cast<float>(i);
        if (i < 0) { i = 0; f = 0.f; }
        if (i >= n_old - 1) { i = n_old - 1; f = 0.f; }
        i0[d] = i;
        a[d] = f;
      }
    };
    taps(nw, ow, inv_fx, x0, ax);
    taps(nh, oh, inv_fy, y0, ay);
    for (auto& ch : channels_) {
      std::vector<double> out(static_cast<size_t>(nw) * nh);
      for (int r = 0; r < nh; ++r) {
        const double* r0 = ch.data() + static_cast<size_t>(y0[r]) * ow;
        const double* r1 = ch.data() + static_cast<size_t>(std::min(y0[r] + 1, oh - 1)) * ow;
        const double wy1 = ay[r], wy0 = 1.f - ay[r];
        for (int c = 0; c < nw; ++c) {
          const int c0 = x0[c], c1 = std::min(x0[c] + 1, ow - 1);
          const double wx1 = ax[c], wx0 = 1.f - ax[c];
          const double top = r0[c0] * wx0 + r0[c1] * wx1, bot = r1[c0] * wx0 + r1[c1] * wx1;
          out[static_cast<size_t>(r) * nw + c] = top * wy0 + bot * wy1;
        }
      }
      ch.swap(out);
    }
    image_size_ = cv::Size(nw, nh);
  }

  // Planar [C][H][W] copy / replacement (used by the facade to cross the C ABI).
  std::vector<double> ToPlanar() const {
    std::vector<double> out;
    out.reserve(static_cast<size_t>(GetNumPixels()) * GetNumChannels());
    for (const auto& c : channels_) out.insert(out.end(), c.begin(), c.end());
    return out;
  }
  void FromPlanar(const std::vector<double>& data, const cv::Size& size, const int num_channels) {
    channels_.clear();
    image_size_ = size;
    const size_t n = static_cast<size_t>(size.area());
    for (int c = 0; c < num_channels; ++c) channels_.emplace_back(data.begin() + c * n, data.begin() + (c + 1) * n);
  }

 private:
  cv::Size image_size_;
  std::vector<std::vector<double>> channels_;
};

}  // namespace super_resolution
