// host_io_test.cpp -- the reference's hyperspectral I/O test cases
// (test/test_hyperspectral_data_loader.cpp:33-110) restated against the drop-in
// loader (super-resolution_amd/host/hyperspectral), on the reference's own data
// files (tests/golden/envi), and the colour-space cases of test/test_image_data.cpp:403-606
// (ChangeColorSpace, InterpolateColorFrom) on the reference's 4 x 4 x 3 literal.  No GPU needed.  argv[1] = golden directory,
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

// ---- colour path (test_image_data.cpp:15-34 literal, :403-606 cases) ----
constexpr double kPixelErrorTolerance = 1.0 / 255.0;
static const double kB[16] = {0.1, 0.2, 0.3, 0.4, 0.15, 0.25, 0.35, 0.45, 0.55, 0.75, 0.85, 0.95, 0.6, 0.65, 0.7, 0.75};
static const double kG[16] = {0.2, 0.3, 0.4, 0.45, 0.1, 0.2, 0.3, 0.4, 0.75, 0.65, 1.0, 1.0, 0.3, 0.35, 0.4, 0.45};
static const double kR[16] = {0.0, 0.05, 0.1, 0.1, 0.0, 0.0, 0.05, 0.1, 0.25, 0.1, 0.2, 0.2, 0.0, 0.05, 0.1, 0.15};

static ImageData ColorImage() {
  ImageData im;
  im.AddChannel(kB, cv::Size(4, 4));
  im.AddChannel(kG, cv::Size(4, 4));
  im.AddChannel(kR, cv::Size(4, 4));
  return im;
}
static bool ChannelNear(const ImageData& a, int ca, const ImageData& b, int cb, double tol) {
  if (a.GetImageSize() != b.GetImageSize()) return false;
  for (int i = 0; i < a.GetNumPixels(); ++i)
    if (std::fabs(a.GetChannelData(ca)[i] - b.GetChannelData(cb)[i]) > tol) return false;
  return true;
}

static void TestChangeColorSpace() {
  ImageData image = ColorImage();
  EXPECT(image.GetNumChannels() == 3 && image.GetSpectralMode() == SPECTRAL_MODE_COLOR_BGR);
  image.ChangeColorSpace(SPECTRAL_MODE_COLOR_YCRCB);
  EXPECT(image.GetNumChannels() == 3 && image.GetSpectralMode() == SPECTRAL_MODE_COLOR_YCRCB);
  // cvtColor's float BGR -> YCrCb on known colours: pixel 0 = (B .1, G .2, R 0)
  const double y0 = 0.299 * 0.0 + 0.587 * 0.2 + 0.114 * 0.1;
  EXPECT(std::fabs(image.GetChannelData(0)[0] - y0) < 1e-6);
  EXPECT(std::fabs(image.GetChannelData(1)[0] - ((0.0 - y0) * 0.713 + 0.5)) < 1e-6);
  EXPECT(std::fabs(image.GetChannelData(2)[0] - ((0.1 - y0) * 0.564 + 0.5)) < 1e-6);
  {  // white and pure red
    const double w[1] = {1.0}, z[1] = {0.0};
    ImageData white; white.AddChannel(w, cv::Size(1, 1)); white.AddChannel(w, cv::Size(1, 1)); white.AddChannel(w, cv::Size(1, 1));
    white.ChangeColorSpace(SPECTRAL_MODE_COLOR_YCRCB);
    EXPECT(std::fabs(white.GetChannelData(0)[0] - 1.0) < 1e-6 && std::fabs(white.GetChannelData(1)[0] - 0.5) < 1e-6 &&
           std::fabs(white.GetChannelData(2)[0] - 0.5) < 1e-6);
    ImageData red; red.AddChannel(z, cv::Size(1, 1)); red.AddChannel(z, cv::Size(1, 1)); red.AddChannel(w, cv::Size(1, 1));
    red.ChangeColorSpace(SPECTRAL_MODE_COLOR_YCRCB);
    EXPECT(std::fabs(red.GetChannelData(0)[0] - 0.299) < 1e-6 && std::fabs(red.GetChannelData(1)[0] - 0.999813) < 1e-5 &&
           std::fabs(red.GetChannelData(2)[0] - 0.331364) < 1e-5);
  }
  // image operations still work on the converted image; converting back restores BGR (:449-487)
  ImageData resized = image;
  resized.ResizeImage(2, INTERPOLATE_NEAREST);
  EXPECT(resized.GetImageSize() == cv::Size(8, 8));
  image.ChangeColorSpace(SPECTRAL_MODE_COLOR_BGR);
  const ImageData original = ColorImage();
  for (int c = 0; c < 3; ++c) EXPECT(ChannelNear(image, c, original, c, kPixelErrorTolerance));
  resized.ChangeColorSpace(SPECTRAL_MODE_COLOR_BGR);
  ImageData original_resized = ColorImage();
  original_resized.ResizeImage(2, INTERPOLATE_NEAREST);
  for (int c = 0; c < 3; ++c) EXPECT(ChannelNear(resized, c, original_resized, c, kPixelErrorTolerance));

  // luminance-only mode (:489-530): one visible channel, chroma hidden and interpolated back
  ImageData reference_ycrcb = ColorImage();
  reference_ycrcb.ChangeColorSpace(SPECTRAL_MODE_COLOR_YCRCB);
  ImageData image_2 = ColorImage();
  image_2.ChangeColorSpace(SPECTRAL_MODE_COLOR_YCRCB, true);
  EXPECT(image_2.GetNumChannels() == 1);
  EXPECT(ChannelNear(image_2, 0, reference_ycrcb, 0, kPixelErrorTolerance));
  EXPECT(image_2.ToPlanar().size() == 16);  // only the luminance crosses the C ABI
  image_2.ResizeImage(2, INTERPOLATE_NEAREST);
  EXPECT(image_2.GetImageSize() == cv::Size(8, 8) && image_2.GetNumChannels() == 1);
  image_2.ChangeColorSpace(SPECTRAL_MODE_COLOR_BGR);
  EXPECT(image_2.GetNumChannels() == 3);
  for (int c = 0; c < 3; ++c) EXPECT(ChannelNear(image_2, c, original_resized, c, 0.15));  // the reference's tolerance
}

static void TestInterpolateColorFrom() {
  ImageData reference_color = ColorImage();
  reference_color.ChangeColorSpace(SPECTRAL_MODE_COLOR_YCRCB);
  ImageData luminance(reference_color.GetChannelData(0), cv::Size(4, 4));
  EXPECT(luminance.GetNumChannels() == 1);
  ImageData luminance_2 = luminance;
  luminance.InterpolateColorFrom(reference_color);
  EXPECT(luminance.GetNumChannels() == 3 && luminance.GetSpectralMode() == SPECTRAL_MODE_COLOR_YCRCB);
  for (int c = 0; c < 3; ++c) EXPECT(ChannelNear(luminance, c, reference_color, c, 1e-12));
  // resized luminance takes bilinearly resized chroma (:573-605)
  luminance_2.ResizeImage(2, INTERPOLATE_LINEAR);
  ImageData reference_resized = reference_color;
  reference_resized.ResizeImage(2, INTERPOLATE_LINEAR);
  EXPECT(luminance_2.GetImageSize() != reference_color.GetImageSize());
  luminance_2.InterpolateColorFrom(reference_color);
  EXPECT(luminance_2.GetNumChannels() == 3);
  for (int c = 0; c < 3; ++c) EXPECT(ChannelNear(luminance_2, c, reference_resized, c, 1e-12));
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
  TestChangeColorSpace();
  TestInterpolateColorFrom();
  std::printf(g_fail ? "HOST IO TESTS FAILED (%d)\n" : "HOST IO TESTS PASSED\n", g_fail);
  return g_fail ? 1 : 0;
}
