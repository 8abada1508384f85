// ObjectiveTerm / ObjectiveFunction and the two terms of the MAP objective
// (src/optimization/objective_function.{h,cpp}:18-72, objective_data_term.{h,cpp},
// objective_irls_regularization_term.{h,cpp}), evaluated by the HIP library through
// the C ABI.  Same contract as the reference: Compute() returns the term's cost and
// ACCUMULATES its gradient into `gradient` (nullptr = cost only);
// ObjectiveFunction::ComputeAllTerms zeroes the gradient, then sums the terms
// (objective_function.cpp:5-20).
//
// These classes exist for callers that assemble an objective term by term (every
// Compute() moves x and the gradient across PCIe); MapSolver::ComputeAllTerms and
// IRLSMapSolver::Solve evaluate the same terms device-resident in one pass.
#pragma once
#include <memory>
#include <vector>

#include "image/image_data.h"
#include "image_model/image_model.h"
#include "optimization/regularizer.h"
#include "util/srmap_host.h"

namespace super_resolution {

class ObjectiveTerm {
 public:
  virtual ~ObjectiveTerm() = default;
  // NOTE: the gradient may be nullptr, in which case it is not computed (objective_function.h:23).
  virtual double Compute(const double* estimated_image_data, double* gradient) const = 0;
};

class ObjectiveFunction {
 public:
  explicit ObjectiveFunction(const int num_parameters) : num_parameters_(num_parameters), num_iterations_completed_(0) {}
  void AddTerm(const std::shared_ptr<ObjectiveTerm> objective_term) { terms_.push_back(objective_term); }
  double ComputeAllTerms(const double* estimated_image_data, double* gradient = nullptr) const {
    if (gradient != nullptr)
      for (int i = 0; i < num_parameters_; ++i) gradient[i] = 0.0;
    double residual_sum = 0.0;
    for (const std::shared_ptr<ObjectiveTerm>& term : terms_) residual_sum += term->Compute(estimated_image_data, gradient);
    return residual_sum;
  }
  void ReportIterationComplete(const double /*residual_sum*/) { num_iterations_completed_++; }
  int GetNumCompletedIterations() const { return num_iterations_completed_; }

 private:
  const int num_parameters_;
  std::vector<std::shared_ptr<ObjectiveTerm>> terms_;
  int num_iterations_completed_;
};

// sum_k ||A_k x - y_k||^2 evaluated at HR resolution on NN-upsampled images, i.e. s^2 * the LR form
// (objective_data_term.cpp:15-116).  `observations` may be the LR frames or -- as MapSolver hands them over in the
// reference (map_solver.cpp:80-85) -- their NN-upsampled HR versions; `image_size` is the HR size.  The term covers
// channels [channel_start, channel_end) of the observations; x and the gradient hold exactly those channels.
class ObjectiveDataTerm : public ObjectiveTerm {
 public:
  ObjectiveDataTerm(const ImageModel& image_model, const std::vector<ImageData>& observations, const int channel_start,
                    const int channel_end, const cv::Size& image_size)
      : num_channels_(channel_end - channel_start), image_size_(image_size) {
    if (observations.empty()) srmap_host::Fail("Cannot super-resolve with 0 low-res images.");
    if (channel_start < 0 || channel_end <= channel_start || channel_end > observations[0].GetNumChannels())
      srmap_host::Fail("ObjectiveDataTerm: invalid channel range");
    srmap_host::ChainParams chain;
    if (!image_model.Canonical(&chain)) srmap_host::Fail("ObjectiveDataTerm needs the [Motion][Blur]Downsampling operator chain");
    const int s = image_model.GetDownsamplingScale();
    chain.frames = static_cast<int>(observations.size());
    if (!chain.shifts_xy.empty() && chain.shifts_xy.size() / 2 < observations.size())
      srmap_host::Fail("fewer motion shifts than observations");
    if (!chain.shifts_xy.empty()) chain.shifts_xy.resize(2 * observations.size());
    problem_ = srmap_host::MakeProblem(chain, image_size.width, image_size.height, num_channels_);
    const int lw = image_size.width / s, lh = image_size.height / s;
    std::vector<double> stack;
    stack.reserve(observations.size() * static_cast<size_t>(num_channels_) * lw * lh);
    for (const ImageData& im : observations) {
      const cv::Size sz = im.GetImageSize();
      const bool upsampled = sz == image_size && s > 1;
      if (!upsampled && !(sz.width == lw && sz.height == lh)) srmap_host::Fail("ObjectiveDataTerm: observation size mismatch");
      for (int c = channel_start; c < channel_end; ++c) {
        const double* src = im.GetChannelData(c);
        for (int i = 0; i < lh; ++i)
          for (int j = 0; j < lw; ++j)
            stack.push_back(upsampled ? src[static_cast<size_t>(i) * s * sz.width + static_cast<size_t>(j) * s]
                                      : src[static_cast<size_t>(i) * lw + j]);
      }
    }
    srmap_host::Check(srmap_set_observations(problem_.get(), stack.data()), "srmap_set_observations");
  }
  double Compute(const double* estimated_image_data, double* gradient) const override {
    if (!estimated_image_data) srmap_host::Fail("CHECK_NOTNULL(estimated_image_data)");
    double cost = 0;
    const size_t n = static_cast<size_t>(image_size_.area()) * num_channels_;
    std::vector<double> g(gradient ? n : 0);
    srmap_host::Check(srmap_eval(problem_.get(), SRMAP_TERM_DATA, estimated_image_data, &cost, gradient ? g.data() : nullptr),
                      "srmap_eval");
    if (gradient)
      for (size_t i = 0; i < n; ++i) gradient[i] += g[i];
    return cost;
  }

 private:
  const int num_channels_;
  const cv::Size image_size_;
  srmap_host::ProblemPtr problem_;
};

// lambda * sum_p w[p] * r[p]^2 with r the regulariser's per-pixel value, gradient through
// Regularizer::ApplyToImageWithDifferentiation with constants lambda * w (objective_irls_regularization_term.cpp:10-58).
// The weight vector is held BY REFERENCE, as in the reference (objective_irls_regularization_term.h:40): the IRLS loop
// rewrites it between rounds.  lambda <= 0: returns 0 and touches nothing (:16-18).
class ObjectiveIRLSRegularizationTerm : public ObjectiveTerm {
 public:
  ObjectiveIRLSRegularizationTerm(const std::shared_ptr<Regularizer> regularizer, const double regularization_parameter,
                                  const std::vector<double>& irls_weights, const int num_channels, const cv::Size& image_size)
      : regularizer_(regularizer), regularization_parameter_(regularization_parameter), irls_weights_(irls_weights),
        num_channels_(num_channels), image_size_(image_size) {
    srmap_host::ChainParams chain;  // scale 1, no motion, no blur: only the regulariser is evaluated
    problem_ = srmap_host::MakeProblem(chain, image_size.width, image_size.height, num_channels);
    int kind = 0, range = 0;
    double decay = 0;
    regularizer->Describe(&kind, &range, &decay);
    if (regularization_parameter > 0.0)
      srmap_host::Check(srmap_add_regularizer(problem_.get(), kind, regularization_parameter, range, decay, nullptr),
                        "srmap_add_regularizer");
  }
  double Compute(const double* estimated_image_data, double* gradient) const override {
    if (regularization_parameter_ <= 0.0) return 0.0;
    if (!estimated_image_data) srmap_host::Fail("CHECK_NOTNULL(estimated_image_data)");
    const size_t n = static_cast<size_t>(image_size_.area()) * num_channels_;
    if (irls_weights_.size() < n) srmap_host::Fail("irls_weights too short");
    srmap_host::Check(srmap_set_irls_weights(problem_.get(), 0, irls_weights_.data()), "srmap_set_irls_weights");
    double cost = 0;
    std::vector<double> g(gradient ? n : 0);
    srmap_host::Check(srmap_eval(problem_.get(), SRMAP_TERM_REG, estimated_image_data, &cost, gradient ? g.data() : nullptr),
                      "srmap_eval");
    if (gradient)
      for (size_t i = 0; i < n; ++i) gradient[i] += g[i];
    return cost;
  }

 private:
  const std::shared_ptr<Regularizer> regularizer_;
  const double regularization_parameter_;
  const std::vector<double>& irls_weights_;
  const int num_channels_;
  const cv::Size image_size_;
  srmap_host::ProblemPtr problem_;
};

}  // namespace super_resolution
