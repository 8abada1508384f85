// host_io_test.cpp -- the reference's hyperspectral I/O test cases
// (test/test_hyperspectral_data_loader.cpp:33-110) restated against the drop-in
// loader (super-resolution_amd/host/hyperspectral), on the reference's own data
// files (tests/golden/envi).  No GPU needed.  argv[1] = golden directory,
// argv[2] = scratch directory.
#include <cmath>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "hyperspectral/hyperspectral_data_loader.h"
#include "util/config_reader.h"

using namespace super_resolution;

static int g_fail = 0;
#define EXPECT(cond)                                                          \
  do {                                                                        \
    if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } \
  } while (0)

constexpr double kPrecisionErrorTolerance = 1e-6;  // float -> double, test_hyperspectral_data_loader.cpp:20

// the golden config names its data file relative to the reference's build tree: same keys, our path
static std::string LocalConfig(const std::string& golden, const std::string& scratch) {
  std::ifstream in(golden + "/test_hs_config.txt");
  std::ofstream out(scratch + "/hs_config_local.txt");
  std::string line;
  while (std::getline(in, line)) {
    if (line.rfind("file", 0) == 0 && line.size() > 4 && line[4] == ' ') line = "file             " + golden + "/example_envi_data";
    out << line << "\n";
  }
  return scratch + "/hs_config_local.txt";
}

static void TestConfigReader(const std::string& golden) {
  util::ConfigurationFileReader r;
  r.SetDelimiter(' ');
  r.ReadFromFile(golden + "/test_hs_config.txt");
  EXPECT(r.GetValue("file") == "../test_data/example_envi_data");  // runs of the delimiter collapse
  EXPECT(r.GetValueAsInt("num_data_bands") == 10);
  EXPECT(r.GetValue("big_endian") == "false");
  EXPECT(!r.HasValue("#"));
  EXPECT(r.GetValueAsInt("no_such_key") == 0);
  EXPECT(r.GetValue("no_such_key").empty());
}

// TEST(HyperspectralDataLoader, ReadHSIHeaderFromFile), :33-49
static void TestReadHeader(const std::string& golden) {
  HSIBinaryDataParameters p;
  p.ReadHeaderFromFile(golden + "/example_envi_header.hdr");
  EXPECT(p.data_format.interleave == HSI_BINARY_INTERLEAVE_BSQ);
  EXPECT(p.data_format.data_type == HSI_DATA_TYPE_FLOAT);
  EXPECT(p.data_format.big_endian == false);
  EXPECT(p.header_offset == 0);
  EXPECT(p.num_data_rows == 11620);
  EXPECT(p.num_data_cols == 11620);
  EXPECT(p.num_data_bands == 1506);
}

// value = band + row / 10 + column / 100 (the comment at :60-66)
static bool ChannelIs(const ImageData& im, int channel, int band, int row0, int col0) {
  const cv::Size sz = im.GetImageSize();
  for (int r = 0; r < sz.height; ++r)
    for (int c = 0; c < sz.width; ++c) {
      const double expect = band + 0.1 * (row0 + r) + 0.01 * (col0 + c);
      if (std::fabs(im.GetChannelData(channel)[r * sz.width + c] - expect) > kPrecisionErrorTolerance) return false;
    }
  return true;
}

// TEST(HyperspectralDataLoader, LoadBinaryData), :52-87
static void TestLoad(const std::string& config) {
  HyperspectralDataLoader loader(config);
  loader.LoadImageFromENVIFile();
  const ImageData image = loader.GetImage();
  EXPECT(image.GetImageSize() == cv::Size(3, 6));
  EXPECT(image.GetNumChannels() == 5);
  const double expected_channel_0[18] = {5.20, 5.21, 5.22, 5.30, 5.31, 5.32, 5.40, 5.41, 5.42,
                                         5.50, 5.51, 5.52, 5.60, 5.61, 5.62, 5.70, 5.71, 5.72};
  const double expected_channel_4[18] = {9.20, 9.21, 9.22, 9.30, 9.31, 9.32, 9.40, 9.41, 9.42,
                                         9.50, 9.51, 9.52, 9.60, 9.61, 9.62, 9.70, 9.71, 9.72};
  for (int i = 0; i < 18; ++i) {
    EXPECT(std::fabs(image.GetChannelData(0)[i] - expected_channel_0[i]) <= kPrecisionErrorTolerance);
    EXPECT(std::fabs(image.GetChannelData(4)[i] - expected_channel_4[i]) <= kPrecisionErrorTolerance);
  }
  for (int c = 0; c < 5; ++c) EXPECT(ChannelIs(image, c, 5 + c, 2, 0));
}

// TEST(HyperspectralDataLoader, SaveBinaryData), :91-110
static void TestSaveRoundTrip(const std::string& config, const std::string& scratch) {
  HyperspectralDataLoader loader1(config);
  loader1.LoadImageFromENVIFile();
  const ImageData original = loader1.GetImage();
  const std::string out_path = scratch + "/hs_data_loader_envi_out";
  HyperspectralDataLoader loader2(out_path);
  loader2.SaveImage(original, HSIBinaryDataFormat());
  HyperspectralDataLoader loader3(out_path + ".config");  // the config file was generated
  loader3.LoadImageFromENVIFile();
  const ImageData saved = loader3.GetImage();
  EXPECT(saved.GetNumChannels() == original.GetNumChannels());
  EXPECT(saved.GetImageSize() == original.GetImageSize());
  for (int c = 0; c < original.GetNumChannels(); ++c)
    for (int i = 0; i < original.GetNumPixels(); ++i)
      EXPECT(std::fabs(saved.GetChannelData(c)[i] - original.GetChannelData(c)[i]) <= kPrecisionErrorTolerance);
  // the generated header reads back with the reader's own (samples = rows) convention
  HSIBinaryDataParameters p;
  p.ReadHeaderFromFile(out_path + ".hdr");
  EXPECT(p.num_data_rows == 6 && p.num_data_cols == 3 && p.num_data_bands == 5);
  // big-endian output, read back through a config that says so
  HyperspectralDataLoader loader4(out_path + "_be");
  HSIBinaryDataFormat be;
  be.big_endian = true;
  loader4.SaveImage(original, be);
  {
    std::ifstream in(out_path + "_be.config");
    std::ofstream out(scratch + "/be_fixed.config");
    std::string line;
    while (std::getline(in, line)) out << (line == "big_endian false" ? "big_endian true" : line) << "\n";
  }
  HyperspectralDataLoader loader5(scratch + "/be_fixed.config");
  loader5.LoadImageFromENVIFile();
  EXPECT(ChannelIs(loader5.GetImage(), 0, 5, 2, 0));
}

// channel shards: two ranks read disjoint band blocks of the configured range
static void TestBandShards(const std::string& config) {
  HyperspectralDataLoader a(config), b(config);
  a.LoadBandRange(0, 3);
  b.LoadBandRange(3, 5);
  EXPECT(a.GetImage().GetNumChannels() == 3 && b.GetImage().GetNumChannels() == 2);
  for (int c = 0; c < 3; ++c) EXPECT(ChannelIs(a.GetImage(), c, 5 + c, 2, 0));
  for (int c = 0; c < 2; ++c) EXPECT(ChannelIs(b.GetImage(), c, 8 + c, 2, 0));
}

int main(int argc, char** argv) {
  if (argc < 3) { std::printf("usage: host_io_test <golden dir> <scratch dir>\n"); return 2; }
  const std::string golden = argv[1], scratch = argv[2];
  const std::string config = LocalConfig(golden, scratch);
  TestConfigReader(golden);
  TestReadHeader(golden);
  TestLoad(config);
  TestSaveRoundTrip(config, scratch);
  TestBandShards(config);
  std::printf(g_fail ? "HOST IO TESTS FAILED (%d)\n" : "HOST IO TESTS PASSED\n", g_fail);
  return g_fail ? 1 : 0;
}
