// The reference's forward-model interface for the MAP path, forwarding to the
// HIP library through the C ABI:
//   DegradationOperator  src/image_model/degradation_operator.h:17-57
//   MotionModule         src/image_model/motion_module.{h,cpp}
//   BlurModule           src/image_model/blur_module.{h,cpp}
//   DownsamplingModule   src/image_model/downsampling_module.{h,cpp}
//   ImageModel, ImageModelParameters, CreateImageModel
//                        src/image_model/image_model.{h,cpp}
//   AdditiveNoiseModule  src/image_model/additive_noise_module.{h,cpp} (data generation: host-side N(0, sigma/255)
//                        per pixel; cv::randn's global stream cannot be reproduced, the generator is seedable)
// Dense operator matrices (GetOperatorMatrix / GetModelMatrix) are test helpers of
// the reference and are not provided.
#pragma once
#include <cstdint>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "image/image_data.h"
#include "motion/motion_shift.h"
#include "util/srmap_host.h"

namespace super_resolution {

namespace srmap_host {
// Runs srmap_apply / srmap_apply_transpose for one chain on an ImageData.
inline void RunChain(const ChainParams& c, ImageData* image, int index, bool transpose) {
  if (!image) Fail("CHECK_NOTNULL(image_data)");
  const cv::Size in = image->GetImageSize();
  const int C = image->GetNumChannels();
  const std::vector<double> src = image->ToPlanar();
  if (!transpose) {
    ProblemPtr p = MakeProblem(c, in.width, in.height, C);
    int lw = 0, lh = 0;
    srmap_problem_lr_size(p.get(), &lw, &lh);
    std::vector<double> dst(static_cast<size_t>(lw) * lh * C);
    Check(srmap_apply(p.get(), index, src.data(), dst.data()), "srmap_apply");
    image->FromPlanar(dst, cv::Size(lw, lh), C);
  } else {
    const cv::Size out(in.width * c.scale, in.height * c.scale);
    ProblemPtr p = MakeProblem(c, out.width, out.height, C);
    std::vector<double> dst(static_cast<size_t>(out.area()) * C);
    Check(srmap_apply_transpose(p.get(), index, src.data(), dst.data()), "srmap_apply_transpose");
    image->FromPlanar(dst, out, C);
  }
}
}  // namespace srmap_host

class DegradationOperator {
 public:
  virtual ~DegradationOperator() = default;
  virtual void ApplyToImage(ImageData* image_data, const int index) const = 0;
  virtual void ApplyTransposeToImage(ImageData* image_data, const int index) const = 0;
  // What this operator contributes to a fused chain.  The reference's interface has no such member
  // (degradation_operator.h:17-57) and its callers subclass DegradationOperator (test/test_image_model.cpp:31-46): the
  // default is "nothing" -- ImageModel::Canonical() does not recognise such an operator, so a model that contains one is
  // applied operator by operator on the host side of the ABI, exactly as the reference's loop does.
  virtual void Describe(srmap_host::ChainParams* chain) const { (void)chain; }
};

class MotionModule : public DegradationOperator {
 public:
  explicit MotionModule(const MotionShiftSequence& sequence) : sequence_(sequence) {}
  void ApplyToImage(ImageData* image_data, const int index) const override {
    sequence_.GetMotionShift(index);
    srmap_host::RunChain(Chain(), image_data, index, false);
  }
  void ApplyTransposeToImage(ImageData* image_data, const int index) const override {
    sequence_.GetMotionShift(index);
    srmap_host::RunChain(Chain(), image_data, index, true);
  }
  void Describe(srmap_host::ChainParams* c) const override { c->shifts_xy = sequence_.Flat(); }

 private:
  srmap_host::ChainParams Chain() const { srmap_host::ChainParams c; Describe(&c); return c; }
  const MotionShiftSequence sequence_;
};

class BlurModule : public DegradationOperator {
 public:
  // blur_radius is the (odd) kernel size, sigma > 0 (blur_module.cpp:13-23).
  BlurModule(const int blur_radius, const double sigma) : blur_radius_(blur_radius), sigma_(sigma) {
    if (blur_radius < 1 || !(sigma > 0.0) || blur_radius % 2 != 1)
      srmap_host::Fail("BlurModule: radius must be odd and >= 1, sigma > 0");
  }
  void ApplyToImage(ImageData* image_data, const int index) const override {
    srmap_host::RunChain(Chain(), image_data, 0, false);
  }
  void ApplyTransposeToImage(ImageData* image_data, const int index) const override {
    srmap_host::RunChain(Chain(), image_data, 0, true);
  }
  void Describe(srmap_host::ChainParams* c) const override { c->blur_ksize = blur_radius_; c->blur_sigma = sigma_; }

 private:
  srmap_host::ChainParams Chain() const { srmap_host::ChainParams c; Describe(&c); return c; }
  const int blur_radius_;
  const double sigma_;
};

class DownsamplingModule : public DegradationOperator {
 public:
  explicit DownsamplingModule(const int scale) : scale_(scale) {
    if (scale < 1) srmap_host::Fail("DownsamplingModule: scale must be >= 1");
  }
  void ApplyToImage(ImageData* image_data, const int index) const override {
    srmap_host::RunChain(Chain(), image_data, 0, false);
  }
  void ApplyTransposeToImage(ImageData* image_data, const int index) const override {
    srmap_host::RunChain(Chain(), image_data, 0, true);
  }
  void Describe(srmap_host::ChainParams* c) const override { c->scale = scale_; }

 private:
  srmap_host::ChainParams Chain() const { srmap_host::ChainParams c; Describe(&c); return c; }
  const int scale_;
};

// additive_noise_module.cpp:19-36: zero-mean Gaussian noise of standard deviation sigma / 255 (the pixels are scaled
// to [0, 1]) on every channel; the transpose is a no-op (:38-44).  Last operator of a data-generation chain only.
class AdditiveNoiseModule : public DegradationOperator {
 public:
  explicit AdditiveNoiseModule(const double sigma, const uint64_t seed = 0x5eedULL) : sigma_(sigma), rng_(seed) {
    if (!(sigma > 0.0)) srmap_host::Fail("Check failed: sigma_ > 0.0");
  }
  void ApplyToImage(ImageData* image_data, const int /*index*/) const override {
    if (!image_data) srmap_host::Fail("CHECK_NOTNULL(image_data)");
    std::normal_distribution<double> gauss(0.0, sigma_ / 255.0);
    for (int c = 0; c < image_data->GetNumChannels(); ++c) {
      double* px = image_data->GetMutableChannelData(c);
      for (int i = 0; i < image_data->GetNumPixels(); ++i) px[i] += gauss(rng_);
    }
  }
  void ApplyTransposeToImage(ImageData* image_data, const int /*index*/) const override {
    if (!image_data) srmap_host::Fail("CHECK_NOTNULL(image_data)");
  }
  void Describe(srmap_host::ChainParams*) const override {}
  void SetSeed(const uint64_t seed) const { rng_.seed(seed); }

 private:
  const double sigma_;
  mutable std::mt19937_64 rng_;
};

struct ImageModelParameters {
  int scale = 2;
  int blur_radius = 0;
  double blur_sigma = 0.0;
  std::string motion_sequence_path = "";
  MotionShiftSequence motion_sequence;
  double noise_sigma = 0.0;
  uint64_t noise_seed = 0x5eedULL;  // not in the reference (cv::randn draws from OpenCV's global generator)
};

class ImageModel {
 public:
  explicit ImageModel(const int downsampling_scale) : downsampling_scale_(downsampling_scale) {
    if (downsampling_scale < 1) srmap_host::Fail("Downsampling scale must be at least 1");
  }
  // image_model.cpp:17-61
  static ImageModel CreateImageModel(const ImageModelParameters& parameters) {
    ImageModel model(parameters.scale);
    if (!parameters.motion_sequence_path.empty() || parameters.motion_sequence.GetNumMotionShifts() > 0) {
      MotionShiftSequence seq = parameters.motion_sequence;
      if (seq.GetNumMotionShifts() == 0) seq.LoadSequenceFromFile(parameters.motion_sequence_path);
      model.AddDegradationOperator(std::make_shared<MotionModule>(seq));
    }
    if (parameters.blur_radius > 0 && parameters.blur_sigma > 0.0)
      model.AddDegradationOperator(std::make_shared<BlurModule>(parameters.blur_radius, parameters.blur_sigma));
    model.AddDegradationOperator(std::make_shared<DownsamplingModule>(parameters.scale));
    if (parameters.noise_sigma > 0.0)  // image_model.cpp:53-58
      model.AddDegradationOperator(std::make_shared<AdditiveNoiseModule>(parameters.noise_sigma, parameters.noise_seed));
    return model;
  }
  void AddDegradationOperator(std::shared_ptr<DegradationOperator> op) { operators_.push_back(op); }
  ImageData ApplyToImage(const ImageData& image_data, const int index) const {
    ImageData degraded = image_data;
    ApplyToImage(&degraded, index);
    return degraded;
  }
  // image_model.cpp:86-91: operators in insertion order.  The canonical chain
  // [Motion][Blur]Downsampling runs as ONE fused device pass.
  void ApplyToImage(ImageData* image_data, const int index) const {
    srmap_host::ChainParams chain;
    // Blur and downsampling ignore the index (blur_module.cpp:25-28, downsampling_module.cpp:19-27): without a
    // MotionModule the fused problem has ONE frame, whatever index the caller passes
    const AdditiveNoiseModule* noise = nullptr;
    if (Canonical(&chain, &noise)) {
      srmap_host::RunChain(chain, image_data, chain.shifts_xy.empty() ? 0 : index, false);
      if (noise) noise->ApplyToImage(image_data, index);
      return;
    }
    for (const auto& op : operators_) op->ApplyToImage(image_data, index);
  }
  // image_model.cpp:93-101: transposes in reverse order.
  void ApplyTransposeToImage(ImageData* image_data, const int index) const {
    srmap_host::ChainParams chain;
    if (Canonical(&chain)) { srmap_host::RunChain(chain, image_data, chain.shifts_xy.empty() ? 0 : index, true); return; }  // a trailing noise module's transpose is a no-op
    for (int i = static_cast<int>(operators_.size()) - 1; i >= 0; --i) operators_[i]->ApplyTransposeToImage(image_data, index);
  }
  int GetDownsamplingScale() const { return downsampling_scale_; }
  // The fused chain description (valid when the operator list is canonical: [Motion][Blur]Downsampling, optionally
  // followed by an AdditiveNoiseModule, which is returned separately and applied on the host).
  bool Canonical(srmap_host::ChainParams* chain, const AdditiveNoiseModule** noise = nullptr) const {
    int stage = 0;  // 0: expect motion/blur/down, 1: after motion, 2: after blur, 3: after down, 5: after noise
    for (const auto& op : operators_) {
      if (const auto* nm = dynamic_cast<const AdditiveNoiseModule*>(op.get())) {
        if (stage != 3) return false;
        stage = 5;
        if (noise) *noise = nm;
        continue;
      }
      const int kind = dynamic_cast<const MotionModule*>(op.get()) ? 1
                       : dynamic_cast<const BlurModule*>(op.get()) ? 2
                       : dynamic_cast<const DownsamplingModule*>(op.get()) ? 3 : 4;
      if (kind == 4 || kind <= stage) return false;
      stage = kind;
      op->Describe(chain);
    }
    return stage == 3 || stage == 5;
  }

 private:
  std::vector<std::shared_ptr<DegradationOperator>> operators_;
  const int downsampling_scale_;
};

}  // namespace super_resolution
