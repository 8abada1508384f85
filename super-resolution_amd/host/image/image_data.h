// ImageData -- the planar double image container of the reference
// (src/image/image_data.h:109-353), reduced to what the MAP gradient path
// touches: channels as contiguous H x W double planes, the pixel-array
// constructor (image_data.cpp:244-265), AddChannel(const double*, Size)
// (:298-308), GetChannelData / GetMutableChannelData (:526-537), deep copies.
// Colour-space conversion, visualisation and OpenCV cv::Mat interop are outside
// the path.
#pragma once
#include <algorithm>
#include <vector>

#include "util/cv_size.h"
#include "util/srmap_host.h"

namespace super_resolution {

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
