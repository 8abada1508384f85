// super_resolution -- the caller of the MAP path (reference:
// src/super_resolution.cpp:38-115 flags, :126-199 SetupAndRunSolver, :269-453
// main), on the drop-in classes: load or generate the LR frames, bilinear
// initial estimate, IRLS-MAP solve on the GPU, optional PSNR against the ground
// truth, save.  Same flag names and defaults.  Not carried over (out of scope,
// DESIGN.md section 7): wavelet-domain solve, L-BFGS / numerical differentiation,
// SSIM, display.
#include <chrono>
#include <cstdio>
#include <iostream>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "apps/app_flags.h"
#include "evaluation/peak_signal_to_noise_ratio.h"
#include "evaluation/structural_similarity.h"
#include "hyperspectral/spectral_pca.h"
#include "image/image_io.h"
#include "image_model/image_model.h"
#include "optimization/irls_map_solver.h"
#include "optimization/regularizer.h"

using namespace super_resolution;

int main(int argc, char** argv) {
  app::Flags flags(argc, argv,
      "super_resolution --data_path=<dir of LR frames | HR image with --generate_lr_images>\n"
      "  [--generate_lr_images] [--noise_sigma=0] [--noise_seed=1] [--number_of_frames=4]\n"
      "  [--ground_truth_image=<path>] [--upsampling_scale=2] [--blur_radius=3] [--blur_sigma=1]\n"
      "  [--motion_sequence_path=<file>] [--optimization_iterations=20] [--split_channels]\n"
      "  [--regularizer=tv|3dtv|btv] [--btv_scale_range=3] [--btv_spatial_decay=0.5]\n"
      "  [--regularization_parameter=0.01] [--solver=cg] [--solver_iterations=50]\n"
      "  [--interpolate_color] [--solve_in_pca_space] [--num_pca_components=0] [--pca_retained_variance=0]\n"
      "  [--evaluators=psnr,ssim] [--result_path=<path>] [--verbose]");
  const std::string data_path = flags.Str("data_path");
  const bool generate_lr_images = flags.Bool("generate_lr_images", false);
  const double noise_sigma = flags.Double("noise_sigma", 0.0);
  const int noise_seed = flags.Int("noise_seed", 1);
  const int number_of_frames = flags.Int("number_of_frames", 4);
  const std::string ground_truth_image = flags.Str("ground_truth_image");
  const int upsampling_scale = flags.Int("upsampling_scale", 2);
  ImageModelParameters model_parameters;
  model_parameters.scale = upsampling_scale;
  model_parameters.blur_radius = flags.Int("blur_radius", 3);
  model_parameters.blur_sigma = flags.Double("blur_sigma", 1.0);
  model_parameters.motion_sequence_path = flags.Str("motion_sequence_path");
  IRLSMapSolverOptions solver_options;
  solver_options.max_num_irls_iterations = flags.Int("optimization_iterations", 20);
  solver_options.max_num_solver_iterations = flags.Int("solver_iterations", 50);
  solver_options.split_channels = flags.Bool("split_channels", false);
  std::string regularizer_name = flags.Str("regularizer", "tv");
  const int btv_scale_range = flags.Int("btv_scale_range", 3);
  const double btv_spatial_decay = flags.Double("btv_spatial_decay", 0.5);
  const double regularization_parameter = flags.Double("regularization_parameter", 0.01);
  const std::string solver_name = flags.Str("solver", "cg");
  const bool interpolate_color = flags.Bool("interpolate_color", false);
  const bool solve_in_pca_space = flags.Bool("solve_in_pca_space", false);
  const int num_pca_components = flags.Int("num_pca_components", 0);
  const double pca_retained_variance = flags.Double("pca_retained_variance", 0.0);
  const std::string evaluators = flags.Str("evaluators");
  const std::string result_path = flags.Str("result_path");
  // not a reference flag: the solver's start x0 as raw little-endian float64 [C][H][W], so that a CPU run of the
  // reference algorithm can start from the IDENTICAL estimate (tests/test_gpu_apps.py compares the two results)
  const std::string save_initial_estimate = flags.Str("save_initial_estimate");
  const bool verbose = flags.Bool("verbose", false);
  flags.RejectUnknown();
  flags.Require("data_path");
  if (solver_name != "cg") std::fprintf(stderr, "WARNING: only the conjugate gradient solver is available; using cg.\n");

  const ImageModel image_model = ImageModel::CreateImageModel(model_parameters);

  ImageData high_res_image;
  std::vector<ImageData> low_res_images;
  if (generate_lr_images) {  // data_path is the ground truth (super_resolution.cpp:285-299)
    high_res_image = util::LoadImage(data_path);
    // the generating model carries the AdditiveNoiseModule (sigma in 0..255 units), the solver's does not
    ImageModelParameters with_noise = model_parameters;
    with_noise.noise_sigma = noise_sigma;
    with_noise.noise_seed = static_cast<uint64_t>(noise_seed);
    const ImageModel image_model_with_noise = ImageModel::CreateImageModel(with_noise);
    for (int i = 0; i < number_of_frames; ++i) low_res_images.push_back(image_model_with_noise.ApplyToImage(high_res_image, i));
  } else {
    low_res_images = util::LoadImages(data_path);
    if (!ground_truth_image.empty()) high_res_image = util::LoadImage(ground_truth_image);
  }
  if (low_res_images.empty()) {
    std::fprintf(stderr, "Check failed: At least one low-resolution image is required for super-resolution.\n");
    return 1;
  }
  const bool has_ground_truth = !ground_truth_image.empty() || generate_lr_images;
  const bool evaluate_results = has_ground_truth && !evaluators.empty();

  // bilinear upsampling of frame 0 in the original spectral space: the evaluation baseline
  ImageData upsampled_image = low_res_images[0];
  upsampled_image.ResizeImage(upsampling_scale, INTERPOLATE_LINEAR);

  // luminance-only colour path (super_resolution.cpp:330-342, 392-395): solve Y, interpolate Cr / Cb
  if (interpolate_color) {
    std::printf("Super-resolving only the luminance channel.\n");
    for (auto& frame : low_res_images) frame.ChangeColorSpace(SPECTRAL_MODE_COLOR_YCRCB, true);
  }

  // spectral PCA (super_resolution.cpp:344-366): solve on the leading components, reconstruct afterwards
  std::unique_ptr<SpectralPCA> spectral_pca;
  if (solve_in_pca_space && !interpolate_color) {
    if (pca_retained_variance > 0.0) spectral_pca.reset(new SpectralPCA(low_res_images, pca_retained_variance));
    else spectral_pca.reset(new SpectralPCA(low_res_images, num_pca_components));
    for (auto& frame : low_res_images) frame = spectral_pca->GetPCAImage(frame);
    std::printf("Super-resolving in PCA space with %d PCA components.\n", low_res_images[0].GetNumChannels());
  }

  ImageData initial_estimate = low_res_images[0];
  initial_estimate.ResizeImage(upsampling_scale, INTERPOLATE_LINEAR);

  if (!save_initial_estimate.empty()) {
    const std::vector<double> planar = initial_estimate.ToPlanar();
    std::FILE* f = std::fopen(save_initial_estimate.c_str(), "wb");
    if (!f || std::fwrite(planar.data(), sizeof(double), planar.size(), f) != planar.size()) {
      std::fprintf(stderr, "ERROR: cannot write '%s'.\n", save_initial_estimate.c_str());
      return 1;
    }
    std::fclose(f);
  }

  IRLSMapSolver solver(solver_options, image_model, low_res_images, verbose);
  if (regularization_parameter > 0.0) {
    std::shared_ptr<Regularizer> regularizer;
    if (regularizer_name == "btv") {
      regularizer = std::make_shared<BilateralTotalVariationRegularizer>(initial_estimate.GetImageSize(),
                                                                         btv_scale_range, btv_spatial_decay);
    } else {
      if (regularizer_name != "tv" && regularizer_name != "3dtv") {
        std::fprintf(stderr, "WARNING: Unknown regularizer option '%s'. Using default Total Variation regularizer.\n",
                     regularizer_name.c_str());
        regularizer_name = "tv";
      }
      auto tv = std::make_shared<TotalVariationRegularizer>(initial_estimate.GetImageSize());
      if (regularizer_name == "3dtv") tv->SetUse3dTotalVariation(true);
      regularizer = tv;
    }
    solver.AddRegularizer(regularizer, regularization_parameter);
  }

  std::printf("Super-resolving from %zu images...\n", low_res_images.size());
  const auto start_time = std::chrono::steady_clock::now();
  ImageData result = solver.Solve(initial_estimate);
  const std::chrono::duration<double> elapsed = std::chrono::steady_clock::now() - start_time;
  std::printf("Done! Finished in %g seconds.\n", elapsed.count());
  if (interpolate_color) {
    result.InterpolateColorFrom(initial_estimate);
    result.ChangeColorSpace(SPECTRAL_MODE_COLOR_BGR);
  }
  if (spectral_pca) result = spectral_pca->ReconstructImage(result);

  if (evaluate_results) {
    size_t pos = 0;
    while (pos <= evaluators.size()) {
      const size_t comma = evaluators.find(',', pos);
      const std::string evaluator = util::TrimString(evaluators.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos));
      if (evaluator == "psnr") {
        const PeakSignalToNoiseRatioEvaluator psnr_evaluator(high_res_image);
        std::cout << "PSNR score on upsampled: " << psnr_evaluator.Evaluate(upsampled_image) << std::endl;
        std::cout << "PSNR score on result:    " << psnr_evaluator.Evaluate(result) << std::endl;
      } else if (evaluator == "ssim") {
        const StructuralSimilarityEvaluator ssim_evaluator(high_res_image);
        std::cout << "SSIM score on upsampled: " << ssim_evaluator.Evaluate(upsampled_image) << std::endl;
        std::cout << "SSIM score on result:    " << ssim_evaluator.Evaluate(result) << std::endl;
      } else if (!evaluator.empty()) {
        std::fprintf(stderr, "ERROR: Unknown/unsupported evaluator '%s'.\n", evaluator.c_str());
      }
      if (comma == std::string::npos) break;
      pos = comma + 1;
    }
  }
  if (!result_path.empty()) util::SaveImage(result, result_path);
  return 0;
}
