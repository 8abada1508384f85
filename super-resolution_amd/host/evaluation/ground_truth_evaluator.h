// GroundTruthEvaluator (src/evaluation/ground_truth_evaluator.h:10-24): evaluators hold the ground truth BY
// REFERENCE, as the reference does -- the caller keeps it alive.
#pragma once
#include "image/image_data.h"

namespace super_resolution {

class GroundTruthEvaluator {
 public:
  explicit GroundTruthEvaluator(const ImageData& ground_truth) : ground_truth_(ground_truth) {}
  virtual ~GroundTruthEvaluator() = default;
  virtual double Evaluate(const ImageData& image) const = 0;

 protected:
  const ImageData& ground_truth_;
};

}  // namespace super_resolution
