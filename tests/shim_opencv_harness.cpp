// tests/shim_opencv_harness.cpp — the cv::Mat / Eigen branch of include/odometry_shim.hpp (-DODOMETRY_SHIM_WITH_OPENCV
// -DODOMETRY_SHIM_WITH_EIGEN, built against tests/stubs: this image has neither library), the branch a maintainer of the reference
// compiles. Checks what a cv::Mat can do that the stand-in Mat cannot: views whose rows lie `step` bytes apart.
//   * ImagePyramid / DepthPyramid of a view == of its continuous clone (the constructor uploads row by row, as cv::GaussianBlur reads);
//   * a header over padded user data likewise;
//   * ComputeDepth rejects non-continuous inputs AND outputs with the reference's message (ref: src/depth_estimate.cpp:259-263);
//   * Affine4f is Eigen's column-major 4x4: Solve returns it, Reset takes it.
// Prints OK or the failed check.
#include <cstdio>
#include <cmath>
#include <sstream>
#include <vector>
#include "../include/odometry_shim.hpp"
using namespace odometry;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

static float tex(int x, int y) { unsigned h = ((unsigned)x * 73856093u) ^ ((unsigned)y * 19349663u); h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; return (float)(h & 255u); }
static bool same(const Mat& a, const Mat& b) {
  if (a.rows != b.rows || a.cols != b.cols) return false;
  for (int y = 0; y < a.rows; y++) for (int x = 0; x < a.cols; x++) if (a.at<float>(y, x) != b.at<float>(y, x)) return false;
  return true;
}
int main() {
  const int rows = 376, cols = 1241;
  // a larger canvas; the frame is a view into its middle: rows lie (cols + 70) * 4 bytes apart
  Mat canvas(rows + 20, cols + 70, PixelType);
  for (int y = 0; y < canvas.rows; y++) for (int x = 0; x < canvas.cols; x++) canvas.at<float>(y, x) = tex(x, y);
  Mat view = canvas(cv::Rect(30, 10, cols, rows));
  CHECK(!view.isContinuous() && view.rows == rows && view.cols == cols && (size_t)view.step == (size_t)(cols + 70) * 4);
  Mat dense = view.clone();
  CHECK(dense.isContinuous() && same(view, dense));
  {
    ImagePyramid pv(4, view, true), pd(4, dense, true);
    for (int l = 0; l < 4; l++) CHECK(same(pv.GetPyramidImage(l), pd.GetPyramidImage(l)));
    CHECK(pv.GetPyramidImage(0).at<float>(100, 200) != dense.at<float>(100, 200));   // smoothed: the kernels did run
    DepthPyramid dv(4, view, false), dd(4, dense, false);
    for (int l = 0; l < 4; l++) CHECK(same(dv.GetPyramidDepth(l), dd.GetPyramidDepth(l)));
    CHECK(dv.GetPyramidDepth(0).at<float>(17, 33) == view.at<float>(17, 33));
  }
  {  // header over padded user data (cv::Mat(rows, cols, type, data, step))
    const size_t pitch = (size_t)(cols + 3) * sizeof(float);
    std::vector<unsigned char> raw(pitch * rows);
    for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) reinterpret_cast<float*>(raw.data() + y * pitch)[x] = dense.at<float>(y, x);
    Mat user(rows, cols, PixelType, raw.data(), pitch);
    CHECK(!user.isContinuous());
    ImagePyramid pu(3, user, false), pd(3, dense, false);
    for (int l = 0; l < 3; l++) CHECK(same(pu.GetPyramidImage(l), pd.GetPyramidImage(l)));
  }
  // stereo pair with disparity 20; ComputeDepth on continuous Mats works, on views it refuses like the reference
  Mat L(rows, cols, PixelType), R(rows, cols, PixelType);
  for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) { L.at<float>(y, x) = tex(x, y); R.at<float>(y, x) = tex(x + 20, y); }
  DepthEstimator de(8.0f, 900.0f, 15.0f, 0.1f, 30.0f, 0.01f, 28.0f, 0.995f, 50, 4, nullptr, nullptr, 386.1448f / 718.856f, 80000);
  Mat val(rows, cols, CV_8U, cv::Scalar(0)), disp(rows, cols, PixelType), dep(rows, cols, PixelType);
  CHECK(de.ComputeDepth(L, R, val, disp, dep) == 0);
  long nval = 0, n20 = 0;
  for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) { nval += val.at<unsigned char>(y, x); n20 += disp.at<float>(y, x) == 20.0f; }
  CHECK(nval > 500 && n20 > 500);
  {
    Mat big(rows + 4, cols + 4, PixelType, cv::Scalar(1.0));
    Mat lv = big(cv::Rect(2, 2, cols, rows));
    std::stringstream cap;
    std::streambuf* keep = std::cout.rdbuf(cap.rdbuf());
    const int st_in = de.ComputeDepth(lv, R, val, disp, dep);                 // non-continuous input
    Mat bigo(rows + 2, cols + 2, PixelType);
    Mat dv = bigo(cv::Rect(1, 1, cols, rows));
    const int st_out = de.ComputeDepth(L, R, val, dv, dep);                   // non-continuous output
    std::cout.rdbuf(keep);
    CHECK(st_in == -1 && st_out == -1);
    const std::string msg = cap.str();
    size_t first = msg.find("The cv::Mat matrix is not continuous in disparity search!");
    CHECK(first != std::string::npos && msg.find("The cv::Mat matrix is not continuous in disparity search!", first + 1) != std::string::npos);
  }
  // poses are Eigen matrices (column-major): a Solve of a frame against itself returns the start pose
  {
    ImagePyramid p(4, L, true);
    DepthPyramid d(4, dep, false);
    Affine4f init = Affine4f::Identity();
    LevenbergMarquardtOptimizer lm(0.01f, 0.995f, {10, 20, 30, 30}, init, nullptr, 1, 28.0f);
    Affine4f T = lm.Solve(p, d, p);
    CHECK(T(3, 3) == 1.0f && std::fabs(T(0, 3)) < 1e-3f && std::fabs(T(2, 3)) < 1e-3f && std::fabs(T(0, 0) - 1.0f) < 1e-4f);
    CHECK(T.data()[15] == T(3, 3) && T.data()[12] == T(0, 3));               // column-major storage
    CHECK(lm.Reset(T, 0.01f) == 0);
  }
  std::printf("OK valid=%ld\n", nval);
  return 0;
}
