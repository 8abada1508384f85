// ImageData -- the planar double image container of the reference
// (src/image/image_data.h:109-353), reduced to what the MAP gradient path
// touches: channels as contiguous H x W double planes, the pixel-array
// constructor (image_data.cpp:244-265), AddChannel(const double*, Size)
// (:298-308), GetChannelData / GetMutableChannelData (:526-537), deep copies.
// Plus the luminance-only colour path of the CLI: BGR <->
// YCrCb (image_data.cpp:366-416) and InterpolateColorFrom (:450-463).
// Visualisation and OpenCV cv::Mat interop stay outside.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "util/cv_size.h"
#include "util/srmap_host.h"

namespace super_resolution {

// ImageData::ResizeImage methods (image_data.h:25-45); ADDITIVE/CUBIC are not on the host path here
enum ResizeInterpolationMethod { INTERPOLATE_NEAREST, INTERPOLATE_LINEAR };

// image_data.h:72-83
enum ImageSpectralMode {
  SPECTRAL_MODE_NONE,
  SPECTRAL_MODE_HYPERSPECTRAL,
  SPECTRAL_MODE_HYPERSPECTRAL_PCA,
  SPECTRAL_MODE_COLOR_BGR,
  SPECTRAL_MODE_COLOR_YCRCB
};

class ImageData {
 public:
  ImageData() {}
  // pixel_values: planar [C][H][W], copied.
  ImageData(const double* pixel_values, const cv::Size& size, const int num_channels = 1)
      : image_size_(size) {
    if (!pixel_values || size.area() <= 0 || num_channels <= 0) srmap_host::Fail("ImageData");
    const size_t n = static_cast<size_t>(size.area());
    for (int c = 0; c < num_channels; ++c) channels_.emplace_back(pixel_values + c * n, pixel_values + (c + 1) * n);
    spectral_mode_ = DefaultSpectralMode(num_channels);
  }
  void AddChannel(const double* pixel_values, const cv::Size& size) {
    if (!channels_.empty() && size != image_size_) srmap_host::Fail("AddChannel: size mismatch");
    image_size_ = size;
    channels_.emplace_back(pixel_values, pixel_values + size.area());
    spectral_mode_ = DefaultSpectralMode(static_cast<int>(channels_.size()));  // image_data.cpp:298-308
  }
  // In YCrCb luminance-only mode the two chroma channels are kept but hidden (image_data.cpp:490-495).
  int GetNumChannels() const {
    if (spectral_mode_ == SPECTRAL_MODE_COLOR_YCRCB && luminance_channel_only_) return 1;
    return static_cast<int>(channels_.size());
  }
  ImageSpectralMode GetSpectralMode() const { return spectral_mode_; }
  void SetSpectralMode(const ImageSpectralMode mode) { spectral_mode_ = mode; }
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
    if (channels_.empty() || !(scale_factor > 0)) srmap_host::Fail("ResizeImage");
    ResizeImage(cv::Size(static_cast<int>(image_size_.width * scale_factor), static_cast<int>(image_size_.height * scale_factor)),
                method);
  }
  void ResizeImage(const cv::Size& new_size, const ResizeInterpolationMethod method = INTERPOLATE_LINEAR) {
    if (channels_.empty()) srmap_host::Fail("Cannot resize an empty image.");
    const int ow = image_size_.width, oh = image_size_.height;
    const int nw = new_size.width, nh = new_size.height;
    if (nw <= 0 || nh <= 0) srmap_host::Fail("ResizeImage: images must have a positive size");
    // hidden chroma planes (luminance-only mode) keep their size until the colour is interpolated back
    if (GetNumChannels() < static_cast<int>(channels_.size()) && chroma_size_.area() == 0) chroma_size_ = image_size_;
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
    const int visible = GetNumChannels();  // hidden chroma keeps its size until the colour is interpolated back
    for (int vc = 0; vc < visible; ++vc) {
      auto& ch = channels_[vc];
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
    for (int c = 0; c < GetNumChannels(); ++c) out.insert(out.end(), channels_[c].begin(), channels_[c].end());
    return out;
  }
  void FromPlanar(const std::vector<double>& data, const cv::Size& size, const int num_channels) {
    channels_.clear();
    image_size_ = size;
    const size_t n = static_cast<size_t>(size.area());
    for (int c = 0; c < num_channels; ++c) channels_.emplace_back(data.begin() + c * n, data.begin() + (c + 1) * n);
    spectral_mode_ = DefaultSpectralMode(num_channels);
    luminance_channel_only_ = false;
  }

  // ChangeColorSpace (image_data.cpp:366-416): BGR <-> YCrCb through cv::cvtColor on a float32 copy.
  // cvtColor's float formulas restated (OpenCV 3.x color.cpp, delta = 0.5; channel order B, G, R in,
  // Y, Cr, Cb out); OpenCV is absent here: parity unpinned beyond the round trip the reference tests.
  //   Y = 0.299 R + 0.587 G + 0.114 B     Cr = (R - Y) 0.713 + 0.5     Cb = (B - Y) 0.564 + 0.5
  //   R = Y + 1.403 (Cr - 0.5)   G = Y - 0.714 (Cr - 0.5) - 0.344 (Cb - 0.5)   B = Y + 1.773 (Cb - 0.5)
  void ChangeColorSpace(const ImageSpectralMode new_color_mode, const bool luminance_only = false) {
    const auto is_color = [](ImageSpectralMode m) { return m == SPECTRAL_MODE_COLOR_BGR || m == SPECTRAL_MODE_COLOR_YCRCB; };
    if (!is_color(spectral_mode_))
      srmap_host::Fail("Cannot convert non-color (monochrome or hyperspectral) images to a different color space.");
    if (!is_color(new_color_mode)) srmap_host::Fail("Invalid color space. new_color_mode must be SPECTRAL_MODE_COLOR_*.");
    if (new_color_mode == spectral_mode_) return;  // already there (the reference warns and returns)
    if (channels_.size() != 3) srmap_host::Fail("colour images have three channels");
    if (new_color_mode == SPECTRAL_MODE_COLOR_YCRCB) {
      luminance_channel_only_ = luminance_only;
      const size_t n = channels_[0].size();
      for (size_t i = 0; i < n; ++i) {
        const float b = static_cast<float>(channels_[0][i]), g = static_cast<float>(channels_[1][i]), r = static_cast<float>(channels_[2][i]);
        const float y = r * 0.299f + g * 0.587f + b * 0.114f;
        channels_[0][i] = y;
        channels_[1][i] = (r - y) * 0.713f + 0.5f;
        channels_[2][i] = (b - y) * 0.564f + 0.5f;
      }
    } else {
      if (luminance_channel_only_) InterpolateColor(channels_, ChromaSize(), &channels_, image_size_);  // chroma up to the luminance size
      chroma_size_ = cv::Size(0, 0);
      const size_t n = channels_[0].size();
      for (size_t i = 0; i < n; ++i) {
        const float y = static_cast<float>(channels_[0][i]), cr = static_cast<float>(channels_[1][i]) - 0.5f, cb = static_cast<float>(channels_[2][i]) - 0.5f;
        channels_[0][i] = y + 1.773f * cb;
        channels_[1][i] = y - 0.714f * cr - 0.344f * cb;
        channels_[2][i] = y + 1.403f * cr;
      }
      luminance_channel_only_ = false;
    }
    spectral_mode_ = new_color_mode;
  }

  // InterpolateColorFrom (image_data.cpp:450-463): this single-channel (luminance) image takes the two
  // chroma channels of `color_image`, bilinearly resized to this image's size when the sizes differ.
  void InterpolateColorFrom(const ImageData& color_image) {
    if (GetNumChannels() != 1) srmap_host::Fail("Color can only be interpolated for single-channel images.");
    if (color_image.channels_.size() != 3) srmap_host::Fail("The given image must have color information for interpolation.");
    channels_.resize(3);
    InterpolateColor(color_image.channels_, color_image.ChromaSize(), &channels_, image_size_);
    spectral_mode_ = color_image.spectral_mode_;
    luminance_channel_only_ = false;
  }

 private:
  static ImageSpectralMode DefaultSpectralMode(const int num_channels) {  // image_data.cpp:36-44
    if (num_channels == 3) return SPECTRAL_MODE_COLOR_BGR;
    if (num_channels > 3) return SPECTRAL_MODE_HYPERSPECTRAL;
    return SPECTRAL_MODE_NONE;
  }
  // image_data.cpp:144-168: channels 1, 2 of `in` (planes of size `from`) resized (INTER_LINEAR) to `target`
  // when the sizes differ
  static void InterpolateColor(const std::vector<std::vector<double>>& in, const cv::Size& from,
                               std::vector<std::vector<double>>* out, const cv::Size& target) {
    for (int i = 1; i < 3; ++i) {
      if (from == target) { const std::vector<double> copy = in[i]; (*out)[i] = copy; continue; }
      ImageData tmp(in[i].data(), from);
      tmp.ResizeImage(target, INTERPOLATE_LINEAR);
      (*out)[i] = tmp.channels_[0];
    }
  }
  // chroma planes keep the size they had when the luminance-only image was first resized
  cv::Size ChromaSize() const { return chroma_size_.area() > 0 ? chroma_size_ : image_size_; }

  cv::Size image_size_;
  cv::Size chroma_size_;  // (0, 0) = same as image_size_
  std::vector<std::vector<double>> channels_;
  ImageSpectralMode spectral_mode_ = SPECTRAL_MODE_NONE;
  bool luminance_channel_only_ = false;
};

}  // namespace super_resolution
