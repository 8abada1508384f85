// facade_test.cpp -- the reference's own MAP-path test cases, restated against
// the drop-in C++ facade (super-resolution_amd/host) that forwards to the HIP
// library through the C ABI.  Run on a GPU box by tests/test_gpu_facade.py.
//   test/test_image_model.cpp:150-225  DownsamplingModule literals
//   test/test_image_model.cpp:350-408  BlurModule literal
//   test/test_tv_regularizer.cpp:61-198, test_btv_regularizer.cpp:21-95
//   test/test_map_solver.cpp:79-199    SmallDataTest (1 / 10 channels / split)
//   test/test_map_solver.cpp:369-469   RegularizationTest (PSNR ordering)
//   test/test_evaluation.cpp:12-47     PSNR literal;  :99-163 SSIM literal
//   src/image_model/additive_noise_module.cpp:19-44  noise statistics, no-op transpose, place in the chain
//   test/test_spectral_pca.cpp:19-137  SpectralPCA literal + reconstruction bounds
//   src/optimization/irls_map_solver.cpp:200-262  the objective assembled term by term (ObjectiveFunction,
//       ObjectiveDataTerm, ObjectiveIRLSRegularizationTerm) equals MapSolver::ComputeAllTerms
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "evaluation/peak_signal_to_noise_ratio.h"
#include "evaluation/structural_similarity.h"
#include "hyperspectral/spectral_pca.h"
#include "image/image_data.h"
#include "image_model/image_model.h"
#include "motion/motion_shift.h"
#include "motion/registration.h"
#include "optimization/irls_map_solver.h"
#include "optimization/regularizer.h"

using namespace super_resolution;

static int g_fail = 0;
#define EXPECT(cond)                                                          \
  do {                                                                        \
    if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } \
  } while (0)

static bool Near(const double* a, const std::vector<double>& b, double tol) {
  for (size_t i = 0; i < b.size(); ++i)
    if (std::fabs(a[i] - b[i]) > tol) return false;
  return true;
}

static const double kSmall[24] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 0, 1, 2, 9, 7, 5, 4, 2, 1, 2, 4, 6, 8, 0, 1};

static void TestDownsamplingModule() {
  const DownsamplingModule down(2);
  ImageData img(kSmall, cv::Size(6, 4));
  down.ApplyToImage(&img, 0);
  EXPECT(img.GetImageSize() == cv::Size(3, 2));
  EXPECT(Near(img.GetChannelData(0), {1, 3, 5, 9, 5, 2}, 0.0));
  ImageData up(kSmall, cv::Size(6, 4));
  down.ApplyTransposeToImage(&up, 0);
  EXPECT(up.GetImageSize() == cv::Size(12, 8));
  const std::vector<double> expected = {
      1, 0, 2, 0, 3, 0, 4, 0, 5, 0, 6, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
      7, 0, 8, 0, 9, 0, 0, 0, 1, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
      9, 0, 7, 0, 5, 0, 4, 0, 2, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
      2, 0, 4, 0, 6, 0, 8, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  EXPECT(Near(up.GetChannelData(0), expected, 0.0));
}

static void TestBlurModule() {
  const BlurModule blur(3, 0.849321);
  const std::vector<double> expected = {1.875,  3.0,  3.125,  2.625,  2.75,   2.4375, 4.5625, 6.25, 5.3125, 3.1875, 2.3125, 1.9375,
                                        5.0,    6.5,  5.75,   3.875,  1.9375, 0.9375, 2.5625, 3.75, 4.3125, 3.6875, 1.6875, 0.5};
  ImageData a(kSmall, cv::Size(6, 4)), b(kSmall, cv::Size(6, 4));
  blur.ApplyToImage(&a, 0);
  blur.ApplyTransposeToImage(&b, 0);
  EXPECT(Near(a.GetChannelData(0), expected, 0.001));
  EXPECT(Near(b.GetChannelData(0), expected, 0.001));
}

static void TestRegularizers() {
  const std::vector<double> tv_image = {0, 0, 1, 0, 1, 3, -3, -1, 0};
  const std::vector<double> tv_expected = {0, 2, 2, 4, 4, 3, 2, 1, 0};
  const TotalVariationRegularizer tv(cv::Size(3, 3));
  std::vector<double> three;
  for (int c = 0; c < 3; ++c) three.insert(three.end(), tv_image.begin(), tv_image.end());
  const std::vector<double> vals = tv.ApplyToImage(three.data(), 3);
  for (int c = 0; c < 3; ++c) EXPECT(Near(vals.data() + 9 * c, tv_expected, 0.0));
  // analytic gradient vs central finite differences of sum(r^2)
  const std::vector<double> ones(9, 1.0);
  const auto vg = tv.ApplyToImageWithDifferentiation(tv_image.data(), ones, 1);
  EXPECT(Near(vg.first.data(), tv_expected, 0.0));
  for (int i = 0; i < 9; ++i) {
    std::vector<double> p = tv_image, m = tv_image;
    p[i] += 1e-6; m[i] -= 1e-6;
    double fp = 0, fm = 0;
    for (double r : tv.ApplyToImage(p.data(), 1)) fp += r * r;
    for (double r : tv.ApplyToImage(m.data(), 1)) fm += r * r;
    EXPECT(std::fabs((fp - fm) / 2e-6 - vg.second[i]) < 1e-4);
  }
  const double btv_image[25] = {0, 0, 1, 2, 1, 0, 1, 3, 2, 3, 5, 4, 3, -2, 1, 4, 6, 9, 3, 0, -3, -1, 0, 6, 0};
  const BilateralTotalVariationRegularizer btv(cv::Size(5, 5), 2, 0.5);
  const std::vector<double> r1 = btv.ApplyToImage(btv_image, 1);
  EXPECT(r1.size() == 25 && r1[0] == 2.8125 && r1[24] == 0.0);
  const BilateralTotalVariationRegularizer btv2(cv::Size(5, 5), 1, 0.25);
  std::vector<double> two(btv_image, btv_image + 25);
  two.insert(two.end(), btv_image, btv_image + 25);
  const std::vector<double> r2 = btv2.ApplyToImage(two.data(), 2);
  EXPECT(r2.size() == 50 && r2[7] == 0.5625 && r2[32] == 0.5625 && r2[24] == 0.0 && r2[49] == 0.0);
  const auto bg = btv.ApplyToImageWithDifferentiation(btv_image, std::vector<double>(25, 0.5), 1);
  EXPECT(bg.first[0] == 2.8125 && bg.first[24] == 0.0 && bg.second.size() == 25);
}

static void TestSmallData(int num_channels, bool split) {
  const double lr_values[4] = {0.4, 0.2, 0.0, 1.0};
  std::vector<ImageData> low_res;
  for (double v : lr_values) {
    ImageData im;
    const std::vector<double> plane(4, v);
    for (int c = 0; c < num_channels; ++c) im.AddChannel(plane.data(), cv::Size(2, 2));
    low_res.push_back(im);
  }
  ImageModelParameters params;
  params.scale = 2;
  params.motion_sequence = MotionShiftSequence({MotionShift(0, 0), MotionShift(-1, 0), MotionShift(0, -1), MotionShift(-1, -1)});
  const ImageModel model = ImageModel::CreateImageModel(params);
  IRLSMapSolverOptions options;
  options.split_channels = split;
  IRLSMapSolver solver(options, model, low_res, false);
  ImageData x0;
  const std::vector<double> zeros(16, 0.0);
  for (int c = 0; c < num_channels; ++c) x0.AddChannel(zeros.data(), cv::Size(4, 4));
  const ImageData result = solver.Solve(x0);
  const std::vector<double> expected = {0.4, 0.2, 0.4, 0.2, 0.0, 1.0, 0.0, 1.0, 0.4, 0.2, 0.4, 0.2, 0.0, 1.0, 0.0, 1.0};
  EXPECT(result.GetNumChannels() == num_channels);
  for (int c = 0; c < num_channels; ++c) EXPECT(Near(result.GetChannelData(c), expected, 0.001));
}

static void TestRegularizationOrdering() {
  // test/test_map_solver.cpp:369-469: 27x27, 3 channels, scale 3, 5 shifts,
  // blur(3, sigma 3), noise sigma 10/255; asserts PSNR(BTV) > PSNR(TV) > PSNR(none)
  // (the reference draws noise with cv::randn; any seeded N(0, sigma) shows the ordering).
  const int S = 27, C = 3;
  std::mt19937 rng(1234);
  std::normal_distribution<double> noise(0.0, 10.0 / 255.0);
  std::vector<double> gt(static_cast<size_t>(S) * S * C);
  for (int c = 0; c < C; ++c)
    for (int r = 0; r < S; ++r)
      for (int q = 0; q < S; ++q)
        gt[(c * S + r) * S + q] = 0.5 + 0.3 * std::sin(0.4 * r + c) * std::cos(0.3 * q) + (((r / 9 + q / 9) % 2) ? 0.15 : -0.15);
  const ImageData ground_truth(gt.data(), cv::Size(S, S), C);
  ImageModelParameters params;
  params.scale = 3;
  params.blur_radius = 3;
  params.blur_sigma = 3.0;
  params.motion_sequence = MotionShiftSequence({MotionShift(0, 0), MotionShift(1, 0), MotionShift(0, 1), MotionShift(1, 1), MotionShift(2, 1)});
  const ImageModel model = ImageModel::CreateImageModel(params);
  std::vector<ImageData> low_res;
  for (int k = 0; k < 5; ++k) {
    ImageData lr = model.ApplyToImage(ground_truth, k);
    for (int c = 0; c < C; ++c) {
      double* d = lr.GetMutableChannelData(c);
      for (int i = 0; i < lr.GetNumPixels(); ++i) d[i] += noise(rng);
    }
    low_res.push_back(lr);
  }
  // initial estimate: nearest-neighbour upsampling of frame 0
  std::vector<double> x0(gt.size());
  for (int c = 0; c < C; ++c)
    for (int r = 0; r < S; ++r)
      for (int q = 0; q < S; ++q) x0[(c * S + r) * S + q] = low_res[0].GetChannelData(c)[(r / 3) * 9 + q / 3];
  const ImageData initial(x0.data(), cv::Size(S, S), C);
  const PeakSignalToNoiseRatioEvaluator psnr(ground_truth);
  IRLSMapSolverOptions options;
  IRLSMapSolver plain(options, model, low_res, false);
  const double p_none = psnr.Evaluate(plain.Solve(initial));
  IRLSMapSolver with_tv(options, model, low_res, false);
  with_tv.AddRegularizer(std::make_shared<TotalVariationRegularizer>(cv::Size(S, S)), 0.01);
  const double p_tv = psnr.Evaluate(with_tv.Solve(initial));
  IRLSMapSolver with_btv(options, model, low_res, false);
  with_btv.AddRegularizer(std::make_shared<BilateralTotalVariationRegularizer>(cv::Size(S, S), 3, 0.5), 0.01);
  const double p_btv = psnr.Evaluate(with_btv.Solve(initial));
  std::printf("PSNR none %.3f  TV %.3f  BTV %.3f\n", p_none, p_tv, p_btv);
  EXPECT(p_tv > p_none);
  EXPECT(p_btv > p_none);
}

static void TestObjectiveTerms() {
  // The reference assembles its objective term by term (irls_map_solver.cpp:212-236): data term over the
  // (NN-upsampled) observations + one IRLS term per regulariser, summed by ObjectiveFunction::ComputeAllTerms, which
  // zeroes the gradient and lets every term ACCUMULATE (objective_function.cpp:5-20).
  const int W = 24, H = 16, C = 2, s = 2;
  std::mt19937 rng(7);
  std::uniform_real_distribution<double> uni(0.0, 1.0);
  std::vector<double> gt(static_cast<size_t>(W) * H * C), x(gt.size());
  for (double& v : gt) v = uni(rng);
  for (double& v : x) v = uni(rng);
  const ImageData ground_truth(gt.data(), cv::Size(W, H), C);
  ImageModelParameters params;
  params.scale = s;
  params.blur_radius = 3;
  params.blur_sigma = 1.0;
  params.motion_sequence = MotionShiftSequence({MotionShift(0, 0), MotionShift(1, 1), MotionShift(0, 1), MotionShift(1, 0)});
  const ImageModel model = ImageModel::CreateImageModel(params);
  std::vector<ImageData> low_res, upsampled;
  for (int k = 0; k < 4; ++k) {
    low_res.push_back(model.ApplyToImage(ground_truth, k));
    // what MapSolver hands to the data term in the reference: NN-upsampled to HR (map_solver.cpp:80-85)
    std::vector<double> up(static_cast<size_t>(W) * H * C);
    for (int c = 0; c < C; ++c)
      for (int r = 0; r < H; ++r)
        for (int q = 0; q < W; ++q) up[(static_cast<size_t>(c) * H + r) * W + q] = low_res[k].GetChannelData(c)[(r / s) * (W / s) + q / s];
    upsampled.push_back(ImageData(up.data(), cv::Size(W, H), C));
  }
  const cv::Size size(W, H);
  const size_t n = gt.size();
  std::vector<double> w_tv(n), w_btv(n);
  for (size_t i = 0; i < n; ++i) { w_tv[i] = 0.5 + uni(rng); w_btv[i] = 0.5 + uni(rng); }
  auto tv = std::make_shared<TotalVariationRegularizer>(size);
  auto btv = std::make_shared<BilateralTotalVariationRegularizer>(size, 3, 0.5);

  // reference value: the fused device path
  IRLSMapSolverOptions options;
  IRLSMapSolver solver(options, model, low_res, false);
  solver.AddRegularizer(tv, 0.03);
  solver.AddRegularizer(btv, 0.02);
  srmap_host::Check(srmap_set_irls_weights(solver.problem(), 0, w_tv.data()), "weights");
  srmap_host::Check(srmap_set_irls_weights(solver.problem(), 1, w_btv.data()), "weights");
  std::vector<double> g_ref(n);
  const double f_ref = solver.ComputeAllTerms(x.data(), g_ref.data());

  for (int variant = 0; variant < 2; ++variant) {  // LR observations / the reference's NN-upsampled ones
    ObjectiveFunction objective(static_cast<int>(n));
    auto data = std::make_shared<ObjectiveDataTerm>(model, variant ? upsampled : low_res, 0, C, size);
    auto t1 = std::make_shared<ObjectiveIRLSRegularizationTerm>(tv, 0.03, w_tv, C, size);
    auto t2 = std::make_shared<ObjectiveIRLSRegularizationTerm>(btv, 0.02, w_btv, C, size);
    auto t0 = std::make_shared<ObjectiveIRLSRegularizationTerm>(btv, 0.0, w_btv, C, size);  // lambda <= 0: skipped
    objective.AddTerm(data);
    objective.AddTerm(t1);
    objective.AddTerm(t2);
    objective.AddTerm(t0);
    std::vector<double> g(n, 123.0);  // ComputeAllTerms must zero it first
    const double f = objective.ComputeAllTerms(x.data(), g.data());
    EXPECT(std::fabs(f - f_ref) <= 1e-12 * std::fabs(f_ref));
    double err = 0;
    for (size_t i = 0; i < n; ++i) err = std::fmax(err, std::fabs(g[i] - g_ref[i]) / std::fmax(1.0, std::fabs(g_ref[i])));
    EXPECT(err <= 1e-11);
    EXPECT(std::fabs(objective.ComputeAllTerms(x.data()) - f) <= 1e-12 * std::fabs(f));  // gradient == nullptr: cost only
    // a term ACCUMULATES into the gradient it is given and returns its own cost
    std::vector<double> acc(n, 1.0), alone(n, 0.0);
    const double c1 = t1->Compute(x.data(), alone.data());
    const double c1b = t1->Compute(x.data(), acc.data());
    EXPECT(c1 == c1b);
    double acc_err = 0;
    for (size_t i = 0; i < n; ++i) acc_err = std::fmax(acc_err, std::fabs(acc[i] - (1.0 + alone[i])));
    EXPECT(acc_err <= 1e-12);
    std::vector<double> untouched(n, 5.0);
    EXPECT(t0->Compute(x.data(), untouched.data()) == 0.0);
    for (size_t i = 0; i < n; ++i) if (untouched[i] != 5.0) { EXPECT(false); break; }
    objective.ReportIterationComplete(f);
    EXPECT(objective.GetNumCompletedIterations() == 1);
  }
  // channel ranges (split_channels: irls_map_solver.cpp:200-210): the data term over channel 1 alone
  {
    auto d1 = std::make_shared<ObjectiveDataTerm>(model, low_res, 1, 2, size);
    std::vector<double> g1(static_cast<size_t>(W) * H, 0.0), gall(n, 0.0);
    const double f1 = d1->Compute(x.data() + static_cast<size_t>(W) * H, g1.data());
    auto dall = std::make_shared<ObjectiveDataTerm>(model, low_res, 0, C, size);
    auto d0 = std::make_shared<ObjectiveDataTerm>(model, low_res, 0, 1, size);
    const double fall = dall->Compute(x.data(), gall.data());
    const double f0 = d0->Compute(x.data(), nullptr);
    EXPECT(std::fabs(f0 + f1 - fall) <= 1e-12 * std::fabs(fall));
    double e = 0;
    for (size_t i = 0; i < g1.size(); ++i) e = std::fmax(e, std::fabs(g1[i] - gall[static_cast<size_t>(W) * H + i]));
    EXPECT(e <= 1e-12);
  }
}

static void TestPsnr() {
  const double gt[16] = {0.0, 0.1, 0.2, 0.3, 0.7, 0.6, 0.5, 0.4, 0.8, 0.9, 1.0, 0.5, 0.4, 0.6, 0.0, 1.0};
  const ImageData ground_truth(gt, cv::Size(4, 4));
  const PeakSignalToNoiseRatioEvaluator psnr(ground_truth);
  EXPECT(std::isinf(psnr.Evaluate(ground_truth)));
  ImageData im(gt, cv::Size(4, 4));
  im.GetMutableChannelData(0)[6] = 0.25;
  im.GetMutableChannelData(0)[15] = 0.5;
  EXPECT(std::fabs(psnr.Evaluate(im) - 17.09269960975831) < 1e-12);
}

// TEST(SpectralPCA, Decomposition), test_spectral_pca.cpp:19-137
static double MaxAbsDiff(const ImageData& a, const ImageData& b) {
  double m = 0;
  for (int c = 0; c < a.GetNumChannels(); ++c)
    for (int i = 0; i < a.GetNumPixels(); ++i) m = std::max(m, std::fabs(a.GetChannelData(c)[i] - b.GetChannelData(c)[i]));
  return m;
}
static void TestSsimAndNoise() {
  // test/test_evaluation.cpp:99-140
  const double gt[4] = {0.5, 0.25, 0.75, 1.0}, im[4] = {0.55, 0.25, 0.7, 1.0};
  ImageData ground_truth(gt, cv::Size(2, 2), 1);
  const StructuralSimilarityEvaluator ssim(ground_truth);
  ImageData test_image(im, cv::Size(2, 2), 1);
  const double expected = 0.991784423266513;
  EXPECT(std::fabs(ssim.Evaluate(test_image) - expected) <= 4e-16 * expected * 4);
  ImageData gt_multi = ground_truth;
  gt_multi.AddChannel(gt, cv::Size(2, 2));
  const StructuralSimilarityEvaluator ssim_multi(gt_multi);
  test_image.AddChannel(im, cv::Size(2, 2));
  EXPECT(std::fabs(ssim_multi.Evaluate(test_image) - expected) <= 4e-16 * expected * 4);
  EXPECT(std::fabs(ssim.Evaluate(ground_truth) - 1.0) <= 1e-15);

  // AdditiveNoiseModule: N(0, sigma / 255) per pixel, reproducible per seed, transpose = no-op, last in the chain
  const int S = 64;
  std::vector<double> flat(static_cast<size_t>(S) * S * 2, 0.5);
  ImageData img(flat.data(), cv::Size(S, S), 2);
  AdditiveNoiseModule noise(10.0, 42);
  ImageData a = img, b = img;
  noise.ApplyToImage(&a, 0);
  noise.SetSeed(42);
  noise.ApplyToImage(&b, 3);
  double sum = 0, sq = 0;
  bool same = true;
  for (int c = 0; c < 2; ++c)
    for (int i = 0; i < S * S; ++i) {
      const double d = a.GetChannelData(c)[i] - 0.5;
      sum += d; sq += d * d;
      if (a.GetChannelData(c)[i] != b.GetChannelData(c)[i]) same = false;
    }
  const double n = 2.0 * S * S, mean = sum / n, sd = std::sqrt(sq / n - mean * mean);
  EXPECT(same);
  EXPECT(std::fabs(mean) < 4 * (10.0 / 255.0) / std::sqrt(n));
  EXPECT(std::fabs(sd - 10.0 / 255.0) < 0.05 * 10.0 / 255.0);
  ImageData t = img;
  noise.ApplyTransposeToImage(&t, 0);
  EXPECT(t.GetChannelData(0)[5] == 0.5);
  ImageModelParameters params;
  params.scale = 2;
  params.noise_sigma = 5.0;
  params.noise_seed = 7;
  const ImageModel noisy = ImageModel::CreateImageModel(params);
  params.noise_sigma = 0.0;
  const ImageModel clean = ImageModel::CreateImageModel(params);
  const ImageData lr_noisy = noisy.ApplyToImage(img, 2), lr_clean = clean.ApplyToImage(img, 2);
  EXPECT(lr_noisy.GetImageSize() == cv::Size(S / 2, S / 2));
  double dev = 0;
  for (int i = 0; i < lr_clean.GetNumPixels(); ++i) dev = std::fmax(dev, std::fabs(lr_noisy.GetChannelData(0)[i] - lr_clean.GetChannelData(0)[i]));
  EXPECT(dev > 0.0 && dev < 6 * 5.0 / 255.0);
}

static void TestSpectralPca() {
  const double ch1[10] = {1.85, 2.05, -0.95, -1.55, -2.55, 2.85, 1.95, 2.75, -2.75, -3.65};
  const double ch2[10] = {2.2175, 2.5425, -1.2075, -1.9575, -3.3825, 3.6425, 2.5925, 3.3175, -3.4825, -4.2825};
  ImageData small_image;
  small_image.AddChannel(ch1, cv::Size(1, 10));
  small_image.AddChannel(ch2, cv::Size(1, 10));
  const SpectralPCA pca_small({small_image});
  const std::vector<double> known1 = {2.88737, 3.266, -1.53633, -2.49680, -4.23402, 4.62459, 3.24237, 4.30858, -4.43722, -5.62453};
  const std::vector<double> known2 = {0.0538, 0.00622, 0.01545, 0.01729, 0.12995, -0.05886, -0.10306, 0.06669, 0.03664, -0.16411};
  const ImageData small_pca = pca_small.GetPCAImage(small_image);
  EXPECT(small_pca.GetNumChannels() == 2);
  EXPECT(Near(small_pca.GetChannelData(0), known1, 1e-5));
  EXPECT(Near(small_pca.GetChannelData(1), known2, 1e-5));
  EXPECT(MaxAbsDiff(pca_small.ReconstructImage(small_pca), small_image) <= 1e-5);

  // bigger image with strongly correlated channels (:62-80; cv::randn replaced by std::normal_distribution)
  ImageData cube;
  const int num_channels = 300;
  const cv::Size size(50, 25);
  std::mt19937_64 rng(12345);
  std::normal_distribution<double> gauss(0.5, 0.1);
  std::vector<double> base(size.area());
  for (int i = 0; i < num_channels; ++i) {
    const double scalar = static_cast<double>(i) / num_channels;
    for (auto& v : base) v = gauss(rng) * scalar;
    cube.AddChannel(base.data(), size);
  }
  const SpectralPCA pca_full({cube});
  EXPECT(MaxAbsDiff(pca_full.ReconstructImage(pca_full.GetPCAImage(cube)), cube) <= 1e-5);
  const SpectralPCA pca_count({cube}, 250);
  const ImageData pca_count_image = pca_count.GetPCAImage(cube);
  EXPECT(pca_count_image.GetNumChannels() == 250);
  EXPECT(MaxAbsDiff(pca_count.ReconstructImage(pca_count_image), cube) <= 0.05);
  const SpectralPCA pca_var({cube}, 0.999);
  const ImageData pca_var_image = pca_var.GetPCAImage(cube);
  EXPECT(pca_var_image.GetNumChannels() < cube.GetNumChannels());
  EXPECT(MaxAbsDiff(pca_var.ReconstructImage(pca_var_image), cube) <= 0.05);
}

// test/test_registration.cpp:27-68: shifts applied with MotionModule are recovered to 0.01 px.  The reference's
// test image is a JPEG (no decoder here): a deterministic textured image stands in for it.
static void TestRegistration() {
  const int W = 320, H = 240;
  std::vector<double> px(static_cast<size_t>(W) * H);
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> uni(0.0, 1.0);
  std::vector<double> coarse(40 * 30);
  for (auto& v : coarse) v = uni(rng);
  for (int r = 0; r < H; ++r)
    for (int c = 0; c < W; ++c) {
      const double u = c / 8.0, v = r / 8.0;  // bilinear blow-up of a random 40 x 30 grid + two sinusoids
      const int u0 = std::min(38, (int)u), v0 = std::min(28, (int)v);
      const double a = u - u0, b = v - v0;
      const double g = (1 - b) * ((1 - a) * coarse[v0 * 40 + u0] + a * coarse[v0 * 40 + u0 + 1]) +
                       b * ((1 - a) * coarse[(v0 + 1) * 40 + u0] + a * coarse[(v0 + 1) * 40 + u0 + 1]);
      px[static_cast<size_t>(r) * W + c] = 0.6 * g + 0.2 + 0.1 * std::sin(0.21 * c) * std::cos(0.17 * r);
    }
  const ImageData original(px.data(), cv::Size(W, H));
  for (int sub = 0; sub < 2; ++sub) {
    std::vector<MotionShift> truth = sub ? std::vector<MotionShift>{MotionShift(0, 0), MotionShift(0.5, -0.25), MotionShift(1.75, 2.5), MotionShift(-3.125, 0.875)}
                                         : std::vector<MotionShift>{MotionShift(0, 0), MotionShift(0, 1), MotionShift(2, 0), MotionShift(5, 5), MotionShift(-5, -1)};
    const MotionShiftSequence truth_seq(truth);
    const MotionModule motion(truth_seq);
    std::vector<ImageData> shifted;
    for (int i = 0; i < truth_seq.GetNumMotionShifts(); ++i) {
      ImageData im = original;
      motion.ApplyToImage(&im, i);
      shifted.push_back(im);
    }
    const MotionShiftSequence got = registration::TranslationalRegistration(shifted);
    EXPECT(got.GetNumMotionShifts() == truth_seq.GetNumMotionShifts());
    const double tol = sub ? 0.05 : 0.01;  // kTranslationEstimateErrorTolerance = 0.01 for the reference's (integer) shifts
    for (int i = 0; i < got.GetNumMotionShifts(); ++i) {
      EXPECT(std::fabs(got[i].dx - truth[i].dx) <= tol);
      EXPECT(std::fabs(got[i].dy - truth[i].dy) <= tol);
    }
  }
  EXPECT(registration::TranslationalRegistration({}).GetNumMotionShifts() == 0);
}

// A caller-defined operator, as the reference's own tests define them (test/test_image_model.cpp:31-46 subclass
// DegradationOperator): it overrides the reference's two members only.  A model that contains it is not a canonical
// chain, so ImageModel applies its operators one by one, in insertion order / reverse order for the transpose
// (image_model.cpp:76-101).
class GainOperator : public DegradationOperator {
 public:
  explicit GainOperator(double gain) : gain_(gain) {}
  void ApplyToImage(ImageData* image, const int) const override { Scale(image); }
  void ApplyTransposeToImage(ImageData* image, const int) const override { Scale(image); }

 private:
  void Scale(ImageData* image) const {
    for (int c = 0; c < image->GetNumChannels(); ++c) {
      double* px = image->GetMutableChannelData(c);
      for (int i = 0; i < image->GetNumPixels(); ++i) px[i] *= gain_;
    }
  }
  const double gain_;
};

static void TestCallerDefinedOperator() {
  ImageModel model(2);
  model.AddDegradationOperator(std::make_shared<GainOperator>(3.0));
  model.AddDegradationOperator(std::make_shared<DownsamplingModule>(2));
  srmap_host::ChainParams chain;
  EXPECT(!model.Canonical(&chain));  // unknown operator: per-operator path
  ImageData img(kSmall, cv::Size(6, 4));
  model.ApplyToImage(&img, 0);
  EXPECT(img.GetImageSize() == cv::Size(3, 2));
  EXPECT(Near(img.GetChannelData(0), {3, 9, 15, 27, 15, 6}, 1e-12));  // 3 x the decimation literal of test_image_model.cpp:188-193
  const double lr[6] = {1, 3, 5, 9, 5, 2};
  ImageData up(lr, cv::Size(3, 2));
  model.ApplyTransposeToImage(&up, 0);  // D^T first, then the gain
  EXPECT(up.GetImageSize() == cv::Size(6, 4));
  EXPECT(Near(up.GetChannelData(0), {3, 0, 9, 0, 15, 0, 0, 0, 0, 0, 0, 0, 27, 0, 15, 0, 6, 0, 0, 0, 0, 0, 0, 0}, 1e-12));
}

int main() {
  TestCallerDefinedOperator();
  TestDownsamplingModule();
  TestBlurModule();
  TestRegularizers();
  TestSmallData(1, false);
  TestSmallData(10, false);
  TestSmallData(10, true);
  TestRegularizationOrdering();
  TestObjectiveTerms();
  TestPsnr();
  TestSsimAndNoise();
  TestSpectralPca();
  TestRegistration();
  std::printf(g_fail ? "FACADE TESTS FAILED (%d)\n" : "FACADE TESTS PASSED\n", g_fail);
  return g_fail ? 1 : 0;
}
