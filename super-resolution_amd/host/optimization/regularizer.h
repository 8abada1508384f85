// Regularizer interface and the two regularisers of the MAP path
// (src/optimization/regularizer.h:13-50, tv_regularizer.{h,cpp},
// btv_regularizer.{h,cpp}) evaluated by the HIP library, bug-compatible with the
// reference gradients.
#pragma once
#include <utility>
#include <vector>

#include "util/cv_size.h"
#include "util/srmap_host.h"

namespace super_resolution {

class Regularizer {
 public:
  explicit Regularizer(const cv::Size& image_size) : image_size_(image_size) {}
  virtual ~Regularizer() = default;
  virtual std::vector<double> ApplyToImage(const double* image_data, const int num_channels) const = 0;
  virtual std::pair<std::vector<double>, std::vector<double>> ApplyToImageWithDifferentiation(
      const double* image_data, const std::vector<double>& gradient_constants, const int num_channels) const = 0;
  // How MapSolver registers this regulariser with the library.
  virtual void Describe(int* kind, int* btv_range, double* btv_decay) const = 0;
  cv::Size GetImageSize() const { return image_size_; }

 protected:
  std::vector<double> Values(const double* x, int num_channels) const {
    if (!x) srmap_host::Fail("CHECK_NOTNULL(image_data)");
    srmap_host::ProblemPtr p = Make(num_channels);
    std::vector<double> v(static_cast<size_t>(image_size_.area()) * num_channels);
    srmap_host::Check(srmap_reg_values(p.get(), 0, x, v.data()), "srmap_reg_values");
    return v;
  }
  std::pair<std::vector<double>, std::vector<double>> ValuesAndGradient(
      const double* x, const std::vector<double>& gc, int num_channels) const {
    if (!x) srmap_host::Fail("CHECK_NOTNULL(image_data)");
    const size_t n = static_cast<size_t>(image_size_.area()) * num_channels;
    if (gc.size() < n) srmap_host::Fail("gradient_constants too short");
    srmap_host::ProblemPtr p = Make(num_channels);
    std::vector<double> v(n), g(n);
    srmap_host::Check(srmap_reg_values_and_gradient(p.get(), 0, x, gc.data(), v.data(), g.data()),
                      "srmap_reg_values_and_gradient");
    return std::make_pair(v, g);
  }
  const cv::Size image_size_;

 private:
  srmap_host::ProblemPtr Make(int num_channels) const {
    srmap_host::ChainParams c;
    srmap_host::ProblemPtr p = srmap_host::MakeProblem(c, image_size_.width, image_size_.height, num_channels);
    int kind = 0, range = 0;
    double decay = 0;
    Describe(&kind, &range, &decay);
    srmap_host::Check(srmap_add_regularizer(p.get(), kind, 1.0, range, decay, nullptr), "srmap_add_regularizer");
    return p;
  }
};

class TotalVariationRegularizer : public Regularizer {
 public:
  using Regularizer::Regularizer;
  std::vector<double> ApplyToImage(const double* image_data, const int num_channels) const override {
    return Values(image_data, num_channels);
  }
  std::pair<std::vector<double>, std::vector<double>> ApplyToImageWithDifferentiation(
      const double* image_data, const std::vector<double>& gradient_constants, const int num_channels) const override {
    return ValuesAndGradient(image_data, gradient_constants, num_channels);
  }
  void SetUse3dTotalVariation(const bool use_3d_total_variation) { use_3d_total_variation_ = use_3d_total_variation; }
  void Describe(int* kind, int* r, double* d) const override {
    *kind = use_3d_total_variation_ ? SRMAP_REG_TV3D : SRMAP_REG_TV; *r = 0; *d = 0;
  }

 private:
  bool use_3d_total_variation_ = false;
};

class BilateralTotalVariationRegularizer : public Regularizer {
 public:
  BilateralTotalVariationRegularizer(const cv::Size& image_size, const int scale_range, const double spatial_decay)
      : Regularizer(image_size), scale_range_(scale_range), spatial_decay_(spatial_decay) {
    if (scale_range < 1) srmap_host::Fail("Range must be at least 1 (1 pixel in each direction).");
    if (!(0 < spatial_decay && spatial_decay <= 1)) srmap_host::Fail("Spatial decay must be between 0 and 1, (0, 1].");
  }
  std::vector<double> ApplyToImage(const double* image_data, const int num_channels) const override {
    return Values(image_data, num_channels);
  }
  std::pair<std::vector<double>, std::vector<double>> ApplyToImageWithDifferentiation(
      const double* image_data, const std::vector<double>& gradient_constants, const int num_channels) const override {
    return ValuesAndGradient(image_data, gradient_constants, num_channels);
  }
  void Describe(int* kind, int* r, double* d) const override { *kind = SRMAP_REG_BTV; *r = scale_range_; *d = spatial_decay_; }

 private:
  const int scale_range_;
  const double spatial_decay_;
};

}  // namespace super_resolution
