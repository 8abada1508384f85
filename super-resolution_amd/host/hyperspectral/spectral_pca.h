// SpectralPCA -- PCA over the spectral (channel) axis of hyperspectral images,
// the step immediately before / after Solve for hyperspectral runs (SURVEY.md 8f,
// row f3; reference src/hyperspectral/spectral_pca.{h,cpp}:30-88, 94-161,
// 165-209, which wraps cv::PCA).  Same class and behaviour:
//   * training data = every (num_pixels / samples_per_image)-th pixel of each
//     image, samples_per_image = min(10 * C / num_images, num_pixels);
//   * cv::PCA(DATA_AS_ROW): mean over the samples, covariance / nsamples,
//     symmetric eigen-decomposition, eigenvalues descending, eigenvectors as
//     rows; `num_pca_bands` (0 = all) or `retained_variance` (the smallest count L
//     whose cumulative eigenvalue share exceeds it, at least 2 -- cv::PCA's
//     computeCumulativeEnergy) components are kept;
//   * GetPCAImage = E (x - mean), ReconstructImage = E^T y + mean, per pixel.
// OpenCV is absent: training runs on the GPU (srmap_channel_pca: row means, centred
// samples, covariance as one DGEMM, rocSOLVER dsyevd); eigenvector signs are fixed
// by "largest-magnitude component positive", which reproduces the one literal the
// reference pins (test_spectral_pca.cpp:19-60); beyond that the signs are
// parity-unpinned (reconstruction and TV/BTV solves are sign-invariant).  The
// per-pixel maps run on the GPU as one dense contraction (srmap_channel_map; on
// device-resident cubes srmap_channel_map_device).
#pragma once
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

#include "image/image_data.h"
#include "util/srmap_host.h"

namespace super_resolution {

class SpectralPCA {
 public:
  SpectralPCA(const std::vector<ImageData>& hyperspectral_images, const int num_pca_bands = 0) {
    Train(hyperspectral_images);
    int keep = num_spectral_bands_;
    if (num_pca_bands > 0) keep = std::min(num_pca_bands, num_spectral_bands_);
    Keep(keep);
  }
  SpectralPCA(const std::vector<ImageData>& hyperspectral_images, const double retained_variance) {
    Train(hyperspectral_images);
    if (!(retained_variance > 0 && retained_variance <= 1)) srmap_host::Fail("retained variance must be in (0, 1]");
    const double total = std::accumulate(eigenvalues_.begin(), eigenvalues_.end(), 0.0);
    int L = 0;
    double cumulative = 0.0;
    for (; L < num_spectral_bands_; ++L) {
      cumulative += eigenvalues_[L];
      if (cumulative / total > retained_variance) break;
    }
    Keep(std::min(std::max(2, L), num_spectral_bands_));
  }

  ImageData GetPCAImage(const ImageData& image_data) const {
    if (image_data.GetNumChannels() != num_spectral_bands_)
      srmap_host::Fail("The input image does not have the correct number of channels.");
    return Map(image_data, basis_, num_pca_bands_, num_spectral_bands_, mean_.data(), nullptr);
  }
  ImageData ReconstructImage(const ImageData& pca_image_data) const {
    if (pca_image_data.GetNumChannels() != num_pca_bands_)
      srmap_host::Fail("The input image does not have the correct number of channels.");
    std::vector<double> bt(static_cast<size_t>(num_spectral_bands_) * num_pca_bands_);
    for (int k = 0; k < num_pca_bands_; ++k)
      for (int c = 0; c < num_spectral_bands_; ++c) bt[static_cast<size_t>(c) * num_pca_bands_ + k] = basis_[static_cast<size_t>(k) * num_spectral_bands_ + c];
    return Map(pca_image_data, bt, num_spectral_bands_, num_pca_bands_, nullptr, mean_.data());
  }
  int GetNumPCABands() const { return num_pca_bands_; }
  const std::vector<double>& GetEigenvalues() const { return eigenvalues_; }

 private:
  static ImageData Map(const ImageData& in, const std::vector<double>& M, int rows_out, int rows_in,
                       const double* offset_in, const double* offset_out) {
    const std::vector<double> x = in.ToPlanar();
    const size_t n = static_cast<size_t>(in.GetNumPixels());
    std::vector<double> y(n * rows_out);
    srmap_host::Check(srmap_channel_map(srmap_host::Context(), rows_out, rows_in, n, M.data(), offset_in, offset_out,
                                        x.data(), y.data()), "srmap_channel_map");
    ImageData out;
    out.FromPlanar(y, in.GetImageSize(), rows_out);
    return out;
  }

  void Keep(int count) {
    num_pca_bands_ = count;
    basis_.resize(static_cast<size_t>(count) * num_spectral_bands_);
  }

  // spectral_pca.cpp:30-88 (sampling) + cv::PCA (mean, covariance / n, eigen, descending)
  void Train(const std::vector<ImageData>& images) {
    if (images.empty()) srmap_host::Fail("At least one image is required to compute the PCA basis.");
    const int C = images[0].GetNumChannels();
    if (C <= 0) srmap_host::Fail("Cannot compute PCA on empty images.");
    const int num_images = static_cast<int>(images.size());
    const int num_pixels = images[0].GetNumPixels();
    int per_image = (C * 10) / num_images;
    if (per_image > num_pixels) per_image = num_pixels;
    if (per_image <= 0) srmap_host::Fail("Too many images for the PCA sampling rule.");
    const int skip = num_pixels / per_image;
    const int ns = num_images * per_image;
    std::vector<double> data(static_cast<size_t>(ns) * C);
    for (int im = 0; im < num_images; ++im) {
      if (images[im].GetNumChannels() != C)
        srmap_host::Fail("Inconsistent number of channels between the given images. Cannot perform PCA.");
      for (int c = 0; c < C; ++c) {
        const double* src = images[im].GetChannelData(c);
        for (int smp = 0; smp < per_image; ++smp) data[(static_cast<size_t>(im) * per_image + smp) * C + c] = src[smp * skip];
      }
    }
    num_spectral_bands_ = C;
    // cv::PCA(data, noArray(), DATA_AS_ROW): mean, covariance / ns, eigenvectors by descending eigenvalue -- on the
    // GPU (srmap_channel_pca: row means, centred samples, covariance as one DGEMM, rocSOLVER dsyevd).  The
    // library wants the samples planar [C][ns]; signs are fixed by "largest-magnitude component positive".
    std::vector<double> planar(static_cast<size_t>(C) * ns);
    for (int i = 0; i < ns; ++i)
      for (int c = 0; c < C; ++c) planar[static_cast<size_t>(c) * ns + i] = data[static_cast<size_t>(i) * C + c];
    mean_.assign(C, 0.0);
    eigenvalues_.assign(C, 0.0);
    basis_.assign(static_cast<size_t>(C) * C, 0.0);
    srmap_host::Check(srmap_channel_pca(srmap_host::Context(), C, static_cast<size_t>(ns), planar.data(), mean_.data(),
                                        eigenvalues_.data(), basis_.data()), "srmap_channel_pca");
    num_pca_bands_ = C;
  }

  int num_spectral_bands_ = 0, num_pca_bands_ = 0;
  std::vector<double> mean_, eigenvalues_;
  std::vector<double> basis_;  // [num_pca_bands][num_spectral_bands], rows = eigenvectors
};

}  // namespace super_resolution
