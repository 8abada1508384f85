// Image files either side of the path for the command-line tools (reference:
// util::LoadImage / LoadImages / SaveImage, src/util/data_loader.cpp).  The
// reference reads whatever OpenCV's imread supports plus ENVI cubes through a
// configuration file; no image codec is available here, so the host tools read
// and write (a) ENVI float32 BSQ cubes (*.config / *.txt, see
// hyperspectral/hyperspectral_data_loader.h) and (b) binary PGM / PPM (P5 / P6,
// 8-bit), normalised to [0, 1] like the reference normalises 8-bit images
// (image_data.cpp:244-265).  A PPM's R, G, B samples become channels 2, 1, 0: ImageData colour images are
// BGR like OpenCV's (SPECTRAL_MODE_COLOR_BGR), which ChangeColorSpace relies on.
#pragma once
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <fstream>
#include <string>
#include <vector>

#include "hyperspectral/hyperspectral_data_loader.h"
#include "image/image_data.h"

namespace super_resolution {
namespace util {

inline std::string GetFileExtension(const std::string& path) {
  const auto pos = path.rfind('.');
  const auto slash = path.rfind('/');
  if (pos == std::string::npos || (slash != std::string::npos && pos < slash)) return "";
  return path.substr(pos + 1);
}
inline bool IsDirectory(const std::string& path) {
  struct stat st;
  return stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
inline bool IsEnviConfig(const std::string& path) {
  const std::string e = GetFileExtension(path);
  return e == "config" || e == "txt";
}

inline ImageData LoadPnm(const std::string& path) {
  std::ifstream in(path, std::ios::binary);
  if (!in.is_open()) hsi_detail::Fatal("Could not open image '" + path + "'.");
  std::string magic;
  in >> magic;
  if (magic != "P5" && magic != "P6") hsi_detail::Fatal("'" + path + "' is not a binary PGM/PPM file.");
  auto next_int = [&]() {
    int c = in.peek();
    while (c == '#' || std::isspace(c)) {
      if (c == '#') { std::string skip; std::getline(in, skip); } else in.get();
      c = in.peek();
    }
    int v = 0;
    in >> v;
    return v;
  };
  const int w = next_int(), h = next_int(), maxv = next_int();
  in.get();  // the single whitespace byte before the raster
  if (w <= 0 || h <= 0 || maxv <= 0 || maxv > 255) hsi_detail::Fatal("Unsupported PGM/PPM header in '" + path + "'.");
  const int nc = magic == "P6" ? 3 : 1;
  std::vector<unsigned char> raw(static_cast<size_t>(w) * h * nc);
  in.read(reinterpret_cast<char*>(raw.data()), static_cast<std::streamsize>(raw.size()));
  if (!in) hsi_detail::Fatal("'" + path + "' is truncated.");
  ImageData image;
  std::vector<double> plane(static_cast<size_t>(w) * h);
  for (int c = 0; c < nc; ++c) {
    const int fc = nc == 3 ? 2 - c : c;  // channel c = B, G, R <- file sample R, G, B
    for (size_t i = 0; i < plane.size(); ++i) plane[i] = raw[i * nc + fc] / static_cast<double>(maxv);
    image.AddChannel(plane.data(), cv::Size(w, h));
  }
  return image;
}

inline void SavePnm(const ImageData& image, const std::string& path) {
  const int nc = image.GetNumChannels();
  if (nc != 1 && nc != 3) hsi_detail::Fatal("PGM/PPM output needs 1 or 3 channels; use an ENVI path for cubes.");
  const int w = image.GetImageSize().width, h = image.GetImageSize().height;
  std::ofstream out(path, std::ios::binary);
  if (!out.is_open()) hsi_detail::Fatal("Could not open '" + path + "' for writing.");
  out << (nc == 3 ? "P6" : "P5") << "\n" << w << " " << h << "\n255\n";
  std::vector<unsigned char> raw(static_cast<size_t>(w) * h * nc);
  for (int c = 0; c < nc; ++c) {
    const double* src = image.GetChannelData(c);
    for (size_t i = 0; i < static_cast<size_t>(w) * h; ++i) {
      const double v = std::min(1.0, std::max(0.0, src[i]));  // saturate like an 8-bit save
      raw[i * nc + (nc == 3 ? 2 - c : c)] = static_cast<unsigned char>(std::lround(v * 255.0));
    }
  }
  out.write(reinterpret_cast<const char*>(raw.data()), static_cast<std::streamsize>(raw.size()));
}

inline ImageData LoadImage(const std::string& path) {
  if (IsEnviConfig(path)) {
    HyperspectralDataLoader loader(path);
    loader.LoadImageFromENVIFile();
    return loader.GetImage();
  }
  return LoadPnm(path);
}

// Every image of a directory, in name order (LR frame i = i-th file).
inline std::vector<ImageData> LoadImages(const std::string& directory) {
  std::vector<std::string> names;
  DIR* d = opendir(directory.c_str());
  if (!d) hsi_detail::Fatal("Could not open directory '" + directory + "'.");
  while (dirent* e = readdir(d)) {
    const std::string n = e->d_name;
    const std::string ext = GetFileExtension(n);
    if (ext == "config" || ext == "pgm" || ext == "ppm") names.push_back(n);
  }
  closedir(d);
  std::sort(names.begin(), names.end(), [](const std::string& a, const std::string& b) {
    return a.size() != b.size() ? a.size() < b.size() : a < b;  // low_res_2 before low_res_10
  });
  std::vector<ImageData> images;
  for (const auto& n : names) images.push_back(LoadImage(directory + "/" + n));
  return images;
}

// ".pgm" / ".ppm" -> PNM; anything else -> ENVI cube at `path` (+ .hdr, .config).
inline void SaveImage(const ImageData& image, const std::string& path) {
  const std::string ext = GetFileExtension(path);
  if (ext == "pgm" || ext == "ppm") { SavePnm(image, path); return; }
  HyperspectralDataLoader(path).SaveImage(image, HSIBinaryDataFormat());
}

}  // namespace util
}  // namespace super_resolution
