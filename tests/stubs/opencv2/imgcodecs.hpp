// tests/stubs/opencv2/imgcodecs.hpp — TEST SCAFFOLDING ONLY (see core.hpp): cv::imread for 8-bit grey PNGs (through the product's own
// std-only decoder, include/odometry_io.hpp, so that the reference's runner compiled against these stubs can read a KITTI-shaped
// directory) and a cv::imwrite that writes nothing. Only what the reference's callers name (ref: run_odometry_kitti_offline.cpp:342,
// 440-458; test_disparity.cpp:92,118).
#pragma once
#include <string>
#include <vector>
#include "core.hpp"
#include "odometry_io.hpp"

namespace cv {
enum ImreadModes { IMREAD_UNCHANGED = -1, IMREAD_GRAYSCALE = 0, IMREAD_COLOR = 1 };
enum ImwriteFlags { IMWRITE_PNG_COMPRESSION = 16 };
inline Mat imread(const std::string& path, int /*flags*/ = IMREAD_COLOR) {
  std::vector<uint8_t> px;
  int w = 0, h = 0;
  if (!odometry::io::read_png_gray8(path, px, w, h)) return Mat();
  Mat m(h, w, CV_8U);
  std::memcpy(m.data, px.data(), (size_t)w * h);
  return m;
}
inline bool imwrite(const std::string&, const Mat&, const std::vector<int>& = std::vector<int>()) { return true; }
}  // namespace cv
