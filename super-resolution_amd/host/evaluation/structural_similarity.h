// StructuralSimilarityEvaluator (src/evaluation/structural_similarity.{h,cpp}): the reference's SSIM is the
// GLOBAL-statistics form over all channels and pixels (no patches -- its own TODO), with c1 = (k1 L)^2,
// c2 = (k2 L)^2; mean and variance of the ground truth are taken at construction (structural_similarity.cpp:52-66).
// Host arithmetic, an evaluation metric outside the gradient path.
#pragma once
#include "evaluation/ground_truth_evaluator.h"
#include "image/image_data.h"
#include "util/srmap_host.h"

namespace super_resolution {

class StructuralSimilarityEvaluator : public GroundTruthEvaluator {
 public:
  StructuralSimilarityEvaluator(const ImageData& ground_truth, const double k1 = 0.01, const double k2 = 0.03,
                                const double image_scale = 1.0)
      : GroundTruthEvaluator(ground_truth) {
    ground_truth_mean_ = Mean(ground_truth);
    ground_truth_variance_ = Covariance(ground_truth, ground_truth_mean_, ground_truth, ground_truth_mean_);
    c1_ = k1 * image_scale;
    c1_ = c1_ * c1_;
    c2_ = k2 * image_scale;
    c2_ = c2_ * c2_;
  }
  // structural_similarity.cpp:68-95
  double Evaluate(const ImageData& image) const override {
    if (image.GetNumChannels() != ground_truth_.GetNumChannels())
      srmap_host::Fail("Check failed: image.GetNumChannels() == ground_truth_.GetNumChannels()");
    ImageData evaluation_image = image;
    if (image.GetImageSize() != ground_truth_.GetImageSize())
      evaluation_image.ResizeImage(ground_truth_.GetImageSize(), INTERPOLATE_LINEAR);
    const double image_mean = Mean(evaluation_image);
    const double image_variance = Covariance(evaluation_image, image_mean, evaluation_image, image_mean);
    const double covariance = Covariance(evaluation_image, image_mean, ground_truth_, ground_truth_mean_);
    const double numerator_1 = 2 * ground_truth_mean_ * image_mean + c1_;
    const double numerator_2 = 2 * covariance + c2_;
    const double denominator_1 = ground_truth_mean_ * ground_truth_mean_ + image_mean * image_mean + c1_;
    const double denominator_2 = ground_truth_variance_ + image_variance + c2_;
    return (numerator_1 * numerator_2) / (denominator_1 * denominator_2);
  }

 private:
  static double Mean(const ImageData& image) {
    const int C = image.GetNumChannels(), n = image.GetNumPixels();
    double sum = 0.0;
    for (int c = 0; c < C; ++c) {
      const double* d = image.GetChannelData(c);
      for (int i = 0; i < n; ++i) sum += d[i];
    }
    return sum / static_cast<double>(C * n);
  }
  static double Covariance(const ImageData& a, const double mean_a, const ImageData& b, const double mean_b) {
    const int C = a.GetNumChannels(), n = a.GetNumPixels();
    double cov = 0.0;
    for (int c = 0; c < C; ++c) {
      const double* da = a.GetChannelData(c);
      const double* db = b.GetChannelData(c);
      for (int i = 0; i < n; ++i) cov += (da[i] - mean_a) * (db[i] - mean_b);
    }
    return cov / static_cast<double>(C * n);
  }
  double ground_truth_mean_, ground_truth_variance_, c1_, c2_;
};

}  // namespace super_resolution
