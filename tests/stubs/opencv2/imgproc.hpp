// tests/stubs/opencv2/imgproc.hpp — TEST SCAFFOLDING ONLY (see core.hpp): the one drawing call the reference's callers name
// (ref: test_disparity.cpp:96), as a no-op.
#pragma once
#include "core.hpp"

namespace cv {
inline void circle(Mat&, Point, int, const Scalar&, int = 1) {}
}  // namespace cv
