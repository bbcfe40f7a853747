// tests/stubs/opencv2/highgui.hpp — TEST SCAFFOLDING ONLY (see core.hpp): the window calls the reference's callers name
// (ref: test_disparity.cpp:100-102), as no-ops.
#pragma once
#include <string>
#include "core.hpp"

namespace cv {
enum WindowFlags { WINDOW_NORMAL = 0, WINDOW_AUTOSIZE = 1 };
inline void namedWindow(const std::string&, int = WINDOW_AUTOSIZE) {}
inline void imshow(const std::string&, const Mat&) {}
inline int waitKey(int = 0) { return -1; }
}  // namespace cv
