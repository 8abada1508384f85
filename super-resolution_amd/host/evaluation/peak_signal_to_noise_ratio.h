// PeakSignalToNoiseRatioEvaluator (src/evaluation/peak_signal_to_noise_ratio.cpp
// :11-54): the parity metric; host arithmetic, MAX = 1.0.
#pragma once
#include <cmath>

#include "image/image_data.h"

namespace super_resolution {

class PeakSignalToNoiseRatioEvaluator {
 public:
  explicit PeakSignalToNoiseRatioEvaluator(const ImageData& ground_truth) : ground_truth_(ground_truth) {}
  double Evaluate(const ImageData& image) const {
    const int num_pixels = image.GetNumPixels(), num_channels = image.GetNumChannels();
    if (num_channels != ground_truth_.GetNumChannels() || image.GetImageSize() != ground_truth_.GetImageSize())
      srmap_host::Fail("Images must have the same size and number of channels to be compared.");
    double ssd = 0.0;
    for (int c = 0; c < num_channels; ++c) {
      const double* a = ground_truth_.GetChannelData(c);
      const double* b = image.GetChannelData(c);
      for (int i = 0; i < num_pixels; ++i) { const double d = a[i] - b[i]; ssd += d * d; }
    }
    const double mse = ssd / static_cast<double>(num_pixels * num_channels);
    return 20.0 * std::log10(1.0) - 10.0 * std::log10(mse);
  }

 private:
  const ImageData ground_truth_;
};

}  // namespace super_resolution
