// generate_data -- low-resolution frames from a high-resolution image through the
// image model (reference: src/generate_data.cpp:83-127, same flag names).  The
// degradation runs on the GPU through the drop-in ImageModel; additive noise is
// drawn on the host with std::normal_distribution (the reference's cv::randn
// stream cannot be reproduced without OpenCV).
#include <cstdio>
#include <random>
#include <string>

#include "apps/app_flags.h"
#include "image/image_io.h"
#include "image_model/image_model.h"

using namespace super_resolution;

int main(int argc, char** argv) {
  app::Flags flags(argc, argv,
      "generate_data --input_image=<ENVI config | .pgm | .ppm> --output_image_dir=<dir>\n"
      "  [--output_image_extension=<pgm|ppm|''(ENVI)>] [--save_as=<path>] [--motion_sequence_path=<file>]\n"
      "  [--blur_radius=0] [--blur_sigma=0] [--noise_sigma=0] [--noise_seed=1]\n"
      "  [--downsampling_scale=2] [--number_of_frames=4]");
  const std::string input_image = flags.Str("input_image");
  const std::string output_dir = flags.Str("output_image_dir");
  std::string extension = flags.Str("output_image_extension");
  const std::string save_as = flags.Str("save_as");
  ImageModelParameters parameters;
  parameters.motion_sequence_path = flags.Str("motion_sequence_path");
  parameters.blur_radius = flags.Int("blur_radius", 0);
  parameters.blur_sigma = flags.Double("blur_sigma", 0.0);
  parameters.noise_sigma = flags.Double("noise_sigma", 0.0);  // 0..255 units (additive_noise_module.cpp:25-26)
  parameters.noise_seed = static_cast<uint64_t>(flags.Int("noise_seed", 1));
  parameters.scale = flags.Int("downsampling_scale", 2);
  const int number_of_frames = flags.Int("number_of_frames", 4);
  flags.RejectUnknown();
  flags.Require("input_image");

  const ImageData image_data = util::LoadImage(input_image);
  if (!save_as.empty()) {  // copy / convert only (generate_data.cpp:94-98)
    util::SaveImage(image_data, save_as);
    return 0;
  }
  flags.Require("output_image_dir");
  const ImageModel image_model = ImageModel::CreateImageModel(parameters);
  if (!extension.empty() && extension[0] != '.') extension = "." + extension;
  for (int i = 0; i < number_of_frames; ++i) {
    const ImageData frame = image_model.ApplyToImage(image_data, i);  // incl. the AdditiveNoiseModule, if any
    const std::string path = output_dir + "/low_res_" + std::to_string(i) + extension;
    util::SaveImage(frame, path);
    std::printf("Generated output image %s\n", path.c_str());
  }
  return 0;
}
