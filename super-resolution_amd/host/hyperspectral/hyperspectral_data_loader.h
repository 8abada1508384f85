// ENVI hyperspectral cubes either side of the MAP path.
// Same classes and behaviour as the reference's loader
// (src/hyperspectral/hyperspectral_data_loader.{h,cpp}:68-118, 120-194, 269-377):
// BSQ, float32, optional byte swap, a key/value configuration file that names
// the data file, its full extent and the [start, end) sub-cube to read; SaveImage
// writes the data, a .hdr and a .config that reads it back.  Reference quirks
// kept on purpose: the header's `samples` is taken as the ROW count and `lines`
// as the COLUMN count (reader and writer agree with each other, not with ENVI),
// and `header_offset` counts ELEMENTS, not bytes.
//
// Beyond the reference: LoadBandRange reads only bands [b0, b1) of the
// configured sub-cube -- one rank's channel shard -- so a multi-GPU run never
// holds the whole cube on one host.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "image/image_data.h"
#include "util/config_reader.h"

namespace super_resolution {

enum HSIDataInterleaveFormat { HSI_BINARY_INTERLEAVE_BSQ };
enum HSIBinaryDataType { HSI_DATA_TYPE_FLOAT };

struct HSIBinaryDataFormat {
  HSIDataInterleaveFormat interleave = HSI_BINARY_INTERLEAVE_BSQ;
  HSIBinaryDataType data_type = HSI_DATA_TYPE_FLOAT;
  bool big_endian = false;
};

namespace hsi_detail {
[[noreturn]] inline void Fatal(const std::string& m) {
  std::fprintf(stderr, "Check failed: %s\n", m.c_str());
  std::abort();
}
inline void Warn(const std::string& m) { std::fprintf(stderr, "WARNING: %s\n", m.c_str()); }
inline bool MachineBigEndian() {
  const uint32_t one = 1;
  unsigned char b[4];
  std::memcpy(b, &one, 4);
  return b[0] != 1;
}
inline float SwapBytes(float v) {
  unsigned char b[4];
  std::memcpy(b, &v, 4);
  const unsigned char r[4] = {b[3], b[2], b[1], b[0]};
  std::memcpy(&v, r, 4);
  return v;
}
}  // namespace hsi_detail

struct HSIBinaryDataParameters {
  HSIBinaryDataParameters() {}

  // ENVI .hdr: "key = value" lines (hyperspectral_data_loader.cpp:214-262).
  void ReadHeaderFromFile(const std::string& header_file_path) {
    util::ConfigurationFileReader r;
    r.SetDelimiter('=');
    r.ReadFromFile(header_file_path);
    if (r.HasValue("interleave") && r.GetValue("interleave") != "bsq")
      hsi_detail::Warn("Unknown/unsupported interleave format: " + r.GetValue("interleave") + ". Using BSQ by default.");
    if (r.HasValue("data type") && r.GetValue("data type") != "4")
      hsi_detail::Warn("Unknown/unsupported data type: " + r.GetValue("data type") + ". Using float by default.");
    if (r.HasValue("byte order")) data_format.big_endian = (r.GetValue("byte order") == "1");
    if (r.HasValue("header offset")) header_offset = r.GetValueAsInt("header offset");
    if (r.HasValue("samples")) num_data_rows = r.GetValueAsInt("samples");  // sic
    if (r.HasValue("lines")) num_data_cols = r.GetValueAsInt("lines");      // sic
    if (r.HasValue("bands")) num_data_bands = r.GetValueAsInt("bands");
  }

  HSIBinaryDataFormat data_format;
  int header_offset = 0;   // in elements
  int num_data_rows = 0;   // extent of the whole file, not of the part read
  int num_data_cols = 0;
  int num_data_bands = 0;
};

class HyperspectralDataLoader {
 public:
  explicit HyperspectralDataLoader(const std::string& file_path) : file_path_(file_path) {}

  // file_path = configuration file (hyperspectral_data_loader.cpp:264-377).
  void LoadImageFromENVIFile() { Load(-1, -1); }

  // Same, restricted to bands [band0, band1) of the configured band range
  // (indices relative to start_band): one rank's channel shard.
  void LoadBandRange(int band0, int band1) {
    if (band0 < 0 || band1 <= band0) hsi_detail::Fatal("Band range must be positive.");
    Load(band0, band1);
  }

  ImageData GetImage() const {
    if (hyperspectral_image_.GetNumChannels() <= 0)
      hsi_detail::Fatal("The hyperspectral image is empty. Make sure to call LoadData() first.");
    return hyperspectral_image_;
  }

  // Data file at file_path, plus file_path.hdr and file_path.config
  // (hyperspectral_data_loader.cpp:120-194).
  void SaveImage(const ImageData& image, const HSIBinaryDataFormat& binary_data_format) const {
    const bool reverse = binary_data_format.big_endian != hsi_detail::MachineBigEndian();
    const int num_rows = image.GetImageSize().height, num_cols = image.GetImageSize().width;
    const int num_bands = image.GetNumChannels();
    {
      std::ofstream out(file_path_, std::ios::binary);
      if (!out.is_open()) hsi_detail::Fatal("ENVI file '" + file_path_ + "' could not be opened for writing.");
      std::vector<float> row(static_cast<size_t>(num_cols));
      for (int band = 0; band < num_bands; ++band) {
        const double* src = image.GetChannelData(band);
        for (int r = 0; r < num_rows; ++r) {
          for (int c = 0; c < num_cols; ++c) {
            float v = static_cast<float>(src[static_cast<size_t>(r) * num_cols + c]);
            row[c] = reverse ? hsi_detail::SwapBytes(v) : v;
          }
          out.write(reinterpret_cast<const char*>(row.data()), static_cast<std::streamsize>(row.size() * sizeof(float)));
        }
      }
    }
    {
      std::ofstream h(file_path_ + ".hdr");
      if (!h.is_open()) hsi_detail::Fatal("Header file '" + file_path_ + ".hdr' could not be opened for writing.");
      h << "ENVI\n"
        << "description = {File generated by HyperspectralDataLoader.}\n"
        << "samples = " << num_rows << "\n"   // sic: rows under `samples`
        << "lines = " << num_cols << "\n"
        << "bands = " << num_bands << "\n"
        << "header offset = 0\n"
        << "file type = ENVI Standard\n"
        << "data type = 4\n"
        << "interleave = bsq\n"
        << "byte order = 0\n";
    }
    {
      std::ofstream c(file_path_ + ".config");
      if (!c.is_open()) hsi_detail::Fatal("Configuration file '" + file_path_ + ".config' could not be opened for writing.");
      c << "# Configuration file for reading '" << file_path_ << "', generated by HyperspectralDataLoader.\n"
        << "file " << file_path_ << "\n"
        << "interleave bsq\n"
        << "data_type float\n"
        << "big_endian false\n"
        << "header_offset 0\n"
        << "num_data_rows " << num_rows << "\n"
        << "num_data_cols " << num_cols << "\n"
        << "num_data_bands " << num_bands << "\n"
        << "start_row 0\n" << "end_row " << num_rows << "\n"
        << "start_col 0\n" << "end_col " << num_cols << "\n"
        << "start_band 0\n" << "end_band " << num_bands << "\n";
    }
  }

 private:
  static int IntOrDie(const util::ConfigurationFileReader& r, const char* key) {
    return std::atoi(r.GetValueOrDie(key).c_str());
  }
  static void Require(bool ok, const char* message) {
    if (!ok) hsi_detail::Fatal(message);
  }

  void Load(int shard_band0, int shard_band1) {
    util::ConfigurationFileReader r;
    r.SetDelimiter(' ');
    r.ReadFromFile(file_path_);
    const std::string data_path = r.GetValueOrDie("file");
    HSIBinaryDataParameters p;
    if (r.GetValueOrDie("interleave") != "bsq")
      hsi_detail::Fatal("Unsupported interleave format: '" + r.GetValue("interleave") + "'.");
    if (r.GetValueOrDie("data_type") != "float")
      hsi_detail::Fatal("Unsupported data type: '" + r.GetValue("data_type") + "'.");
    p.data_format.big_endian = (r.GetValueOrDie("big_endian") == "true");
    p.header_offset = IntOrDie(r, "header_offset");
    Require(p.header_offset >= 0, "Header offset must be non-negative.");
    p.num_data_rows = IntOrDie(r, "num_data_rows");
    Require(p.num_data_rows > 0, "Number of data rows must be positive.");
    p.num_data_cols = IntOrDie(r, "num_data_cols");
    Require(p.num_data_cols > 0, "Number of data cols must be positive.");
    p.num_data_bands = IntOrDie(r, "num_data_bands");
    Require(p.num_data_bands > 0, "Number of data bands must be positive.");
    const int start_row = IntOrDie(r, "start_row"), end_row = IntOrDie(r, "end_row");
    Require(start_row >= 0, "Start row index cannot be negative.");
    Require(start_row < p.num_data_rows, "Start row index is out of bounds.");
    Require(end_row > 0, "End row index must be positive.");
    Require(end_row <= p.num_data_rows, "End row index is out of bounds.");
    Require(end_row - start_row > 0, "Row range must be positive.");
    const int start_col = IntOrDie(r, "start_col"), end_col = IntOrDie(r, "end_col");
    Require(start_col >= 0, "Start column index cannot be negative.");
    Require(start_col < p.num_data_cols, "Start column index is out of bounds.");
    Require(end_col > 0, "End column index must be positive.");
    Require(end_col <= p.num_data_cols, "End column index is out of bounds.");
    Require(end_col - start_col > 0, "Column range must be positive.");
    int start_band = IntOrDie(r, "start_band"), end_band = IntOrDie(r, "end_band");
    Require(start_band >= 0, "Start band index cannot be negative.");
    Require(start_band < p.num_data_bands, "Start band index is out of bounds.");
    Require(end_band > 0, "End band index must be positive.");
    Require(end_band <= p.num_data_bands, "End band index is out of bounds.");
    Require(end_band - start_band > 0, "Band range must be positive.");
    if (shard_band0 >= 0) {  // channel shard inside the configured range
      Require(start_band + shard_band1 <= end_band, "Band shard exceeds the configured band range.");
      end_band = start_band + shard_band1;
      start_band += shard_band0;
    }

    std::ifstream in(data_path, std::ios::binary);
    if (!in.is_open())
      hsi_detail::Fatal("The HSI file path '" + data_path + "' specified in configuration file '" + file_path_ +
                        "' is not a valid ENVI file.");
    const bool reverse = p.data_format.big_endian != hsi_detail::MachineBigEndian();
    const int rows = end_row - start_row, cols = end_col - start_col;
    const int64_t plane = static_cast<int64_t>(p.num_data_rows) * p.num_data_cols;
    hyperspectral_image_ = ImageData();
    std::vector<float> line(static_cast<size_t>(cols));
    std::vector<double> channel(static_cast<size_t>(rows) * cols);
    for (int band = start_band; band < end_band; ++band) {
      for (int row = start_row; row < end_row; ++row) {
        const int64_t first = p.header_offset + band * plane + static_cast<int64_t>(row) * p.num_data_cols + start_col;
        in.seekg(first * static_cast<int64_t>(sizeof(float)));
        in.read(reinterpret_cast<char*>(line.data()), static_cast<std::streamsize>(line.size() * sizeof(float)));
        if (!in) hsi_detail::Fatal("The HSI file '" + data_path + "' is shorter than its configuration says.");
        for (int c = 0; c < cols; ++c)
          channel[static_cast<size_t>(row - start_row) * cols + c] =
              static_cast<double>(reverse ? hsi_detail::SwapBytes(line[c]) : line[c]);
      }
      hyperspectral_image_.AddChannel(channel.data(), cv::Size(cols, rows));
    }
  }

  const std::string file_path_;
  ImageData hyperspectral_image_;
};

}  // namespace super_resolution
