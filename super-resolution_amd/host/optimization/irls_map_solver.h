// Solver / MapSolver / IRLSMapSolver and their option structs
// (src/optimization/solver.h:14-43, map_solver.{h,cpp}, irls_map_solver.{h,cpp}),
// evaluated on the GPU through the C ABI; the term-by-term classes
// (ObjectiveFunction / ObjectiveTerm / ObjectiveDataTerm /
// ObjectiveIRLSRegularizationTerm) live in optimization/objective_function.h.
// Only the CG solver with analytic
// differentiation exists (the reference's L-BFGS / numeric-difference variants
// are alternatives outside the path; the enum is kept for source parity).
#pragma once
#include <cmath>
#include <iostream>
#include <limits>
#include <memory>
#include <utility>
#include <vector>

#include "image/image_data.h"
#include "image_model/image_model.h"
#include "optimization/objective_function.h"
#include "optimization/regularizer.h"
#include "util/srmap_host.h"

namespace super_resolution {

enum LeastSquaresSolver { CG_SOLVER, LBFGS_SOLVER };

struct MapSolverOptions {
  MapSolverOptions() {}
  virtual ~MapSolverOptions() = default;
  virtual void AdjustThresholdsAdaptively(const int num_parameters, const double regularization_parameter_sum) {
    const double threshold_scale = num_parameters * regularization_parameter_sum;
    if (threshold_scale < 1.0) return;
    gradient_norm_threshold *= threshold_scale;
    cost_decrease_threshold *= threshold_scale;
    parameter_variation_threshold *= threshold_scale;
  }
  virtual void PrintSolverOptions() const {
    std::cout << "  Least squares solver:                conjugate gradient (analytical differentiation)\n"
              << "  Threshold 1 (gradient norm):         " << gradient_norm_threshold << "\n"
              << "  Threshold 2 (cost decrease):         " << cost_decrease_threshold << "\n"
              << "  Threshold 3 (parameter variation):   " << parameter_variation_threshold << std::endl;
  }
  LeastSquaresSolver least_squares_solver = CG_SOLVER;
  int num_lbfgs_hessian_corrections = 5;
  int max_num_solver_iterations = 50;
  double gradient_norm_threshold = 1.0e-6;
  double cost_decrease_threshold = 1.0e-6;
  double parameter_variation_threshold = 1.0e-6;
  bool use_numerical_differentiation = false;
  double numerical_differentiation_step = 1.0e-6;
  bool split_channels = false;
};

struct IRLSMapSolverOptions : public MapSolverOptions {
  IRLSMapSolverOptions() {}
  void AdjustThresholdsAdaptively(const int num_parameters, const double regularization_parameter_sum) override {
    const double threshold_scale = num_parameters * regularization_parameter_sum;
    if (threshold_scale < 1.0) return;
    MapSolverOptions::AdjustThresholdsAdaptively(num_parameters, regularization_parameter_sum);
    irls_cost_difference_threshold *= threshold_scale;
  }
  int max_num_irls_iterations = 20;
  double irls_cost_difference_threshold = 1.0e-5;
};

class Solver {
 public:
  explicit Solver(const ImageModel& image_model, const bool verbose = true)
      : image_model_(image_model), is_verbose_(verbose) {}
  virtual ~Solver() = default;
  virtual ImageData Solve(const ImageData& initial_estimate) = 0;
  virtual void Stfu() { is_verbose_ = false; }
  virtual bool IsVerbose() const { return is_verbose_; }

 protected:
  const ImageModel& image_model_;  // the caller keeps the model alive (solver.h:37)
  bool is_verbose_ = true;
};

class MapSolver : public Solver {
 public:
  MapSolver(const ImageModel& image_model, const std::vector<ImageData>& low_res_images,
            const bool print_solver_output = true)
      : Solver(image_model, print_solver_output) {
    if (low_res_images.empty()) srmap_host::Fail("Cannot super-resolve with 0 low-res images.");
    num_channels_ = low_res_images[0].GetNumChannels();
    for (const ImageData& im : low_res_images)
      if (im.GetNumChannels() != num_channels_) srmap_host::Fail("Image channel counts do not match up.");
    const int s = image_model_.GetDownsamplingScale();
    const cv::Size lr = low_res_images[0].GetImageSize();
    image_size_ = cv::Size(lr.width * s, lr.height * s);
    // The library keeps the observations at LR resolution (the reference stores
    // them NN-upsampled, map_solver.cpp:80-85; algebraically identical).
    srmap_host::ChainParams chain;
    if (!image_model_.Canonical(&chain))
      srmap_host::Fail("MapSolver needs the [Motion][Blur]Downsampling operator chain");
    chain.frames = static_cast<int>(low_res_images.size());
    if (!chain.shifts_xy.empty() && chain.shifts_xy.size() / 2 < low_res_images.size())
      srmap_host::Fail("fewer motion shifts than observations");
    if (!chain.shifts_xy.empty()) chain.shifts_xy.resize(2 * low_res_images.size());
    problem_ = srmap_host::MakeProblem(chain, image_size_.width, image_size_.height, num_channels_);
    std::vector<double> stack;
    for (const ImageData& im : low_res_images) {
      if (im.GetImageSize() != lr) srmap_host::Fail("observation sizes differ");
      const std::vector<double> planar = im.ToPlanar();
      stack.insert(stack.end(), planar.begin(), planar.end());
    }
    num_images_ = static_cast<int>(low_res_images.size());
    srmap_host::Check(srmap_set_observations(problem_.get(), stack.data()), "srmap_set_observations");
  }
  virtual void AddRegularizer(std::shared_ptr<Regularizer> regularizer, const double regularization_parameter) {
    regularizers_.push_back(std::make_pair(regularizer, regularization_parameter));
    int kind = 0, range = 0;
    double decay = 0;
    regularizer->Describe(&kind, &range, &decay);
    srmap_host::Check(srmap_add_regularizer(problem_.get(), kind, regularization_parameter, range, decay, nullptr),
                      "srmap_add_regularizer");
  }
  int GetNumPixels() const { return image_size_.width * image_size_.height; }
  cv::Size GetImageSize() const { return image_size_; }
  int GetNumChannels() const { return num_channels_; }
  int GetNumImages() const { return num_images_; }
  int GetNumDataPoints() const {
    const long n = static_cast<long>(GetNumPixels()) * GetNumChannels();
    if (n > std::numeric_limits<int>::max()) srmap_host::Fail("Number of data points exceeds maximum size.");
    return static_cast<int>(n);
  }
  double GetRegularizationParameterSum() const {
    double sum = 0.0;
    for (const auto& r : regularizers_) sum += r.second;
    return sum;
  }
  // ObjectiveFunction::ComputeAllTerms on the current term set
  // (objective_function.cpp:5-20); gradient may be nullptr.
  double ComputeAllTerms(const double* estimated_image_data, double* gradient = nullptr) const {
    double cost = 0;
    srmap_host::Check(srmap_eval(problem_.get(), SRMAP_TERM_ALL, estimated_image_data, &cost, gradient), "srmap_eval");
    return cost;
  }
  srmap_problem* problem() const { return problem_.get(); }

 protected:
  std::vector<std::pair<std::shared_ptr<Regularizer>, double>> regularizers_;
  srmap_host::ProblemPtr problem_;

 private:
  cv::Size image_size_;
  int num_channels_ = 0;
  int num_images_ = 0;
};

class IRLSMapSolver : public MapSolver {
 public:
  IRLSMapSolver(const IRLSMapSolverOptions& solver_options, const ImageModel& image_model,
                const std::vector<ImageData>& low_res_images, const bool print_solver_output = true)
      : MapSolver(image_model, low_res_images, print_solver_output), solver_options_(solver_options) {}

  // irls_map_solver.cpp:192-265
  ImageData Solve(const ImageData& initial_estimate) override {
    if (initial_estimate.GetNumPixels() != GetNumPixels() || initial_estimate.GetNumChannels() != GetNumChannels() ||
        initial_estimate.GetImageSize() != GetImageSize())
      srmap_host::Fail("initial estimate does not match the HR geometry");
    srmap_irls_options o;
    srmap_irls_options_default(&o);
    o.max_num_solver_iterations = solver_options_.max_num_solver_iterations;
    o.gradient_norm_threshold = solver_options_.gradient_norm_threshold;
    o.cost_decrease_threshold = solver_options_.cost_decrease_threshold;
    o.parameter_variation_threshold = solver_options_.parameter_variation_threshold;
    o.split_channels = solver_options_.split_channels ? 1 : 0;
    o.max_num_irls_iterations = solver_options_.max_num_irls_iterations;
    o.irls_cost_difference_threshold = solver_options_.irls_cost_difference_threshold;
    if (solver_options_.least_squares_solver != CG_SOLVER || solver_options_.use_numerical_differentiation)
      srmap_host::Fail("only CG with analytical differentiation is provided");
    const std::vector<double> x0 = initial_estimate.ToPlanar();
    std::vector<double> x(x0.size());
    srmap_host::Check(srmap_solve(problem_.get(), &o, x0.data(), x.data(), &report_), "srmap_solve");
    if (IsVerbose())
      std::cout << "IRLSMapSolver: " << report_.irls_rounds << " IRLS rounds, " << report_.cg_iterations
                << " CG iterations, " << report_.evaluations << " cost+gradient evaluations, final cost "
                << report_.final_cost << std::endl;
    ImageData result;
    result.FromPlanar(x, GetImageSize(), GetNumChannels());
    return result;
  }
  const srmap_solve_report& GetReport() const { return report_; }

 private:
  const IRLSMapSolverOptions solver_options_;
  srmap_solve_report report_ = {};
};

}  // namespace super_resolution
