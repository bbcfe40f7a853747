// tests/shim_mat_harness.cpp — the stand-in odometry::Mat of include/odometry_shim.hpp and its device mirror: writes through
// any header copy invalidate the mirror, device-side results reach host code on first look. Prints OK or the failed check.
#include <cstdio>
#include <csignal>
#include <execinfo.h>
#include <unistd.h>
#include "../include/odometry_shim.hpp"
using namespace odometry;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)
static void on_segv(int) { void* bt[48]; const int n = backtrace(bt, 48); backtrace_symbols_fd(bt, n, 2); _exit(139); }
#define MARK(s) std::fprintf(stderr, "[harness] %s\n", s)
int main() {
  std::signal(SIGSEGV, on_segv);
  const int rows = 376, cols = 1241;
  Mat a(rows, cols, PixelType);
  for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) a.at<float>(y, x) = (float)((x * 7 + y * 13) % 251);
  const Mat& ca = a;
  MARK("filled");
  {
    ImagePyramid p1(1, ca, false);                    // uploads: the mirror is now current
    CHECK(p1.GetPyramidImage(0).at<float>(5, 7) == ca.at<float>(5, 7));
    MARK("p1 ok");
    a.at<float>(5, 7) = 1234.0f;                      // non-const access: mirror invalidated
    ImagePyramid p2(1, ca, false);
    CHECK(p2.GetPyramidImage(0).at<float>(5, 7) == 1234.0f);
    MARK("p2 ok");
    Mat b = a;                                        // header copy shares pixels and mirror
    b.ptr<float>(9)[3] = -5.0f;
    ImagePyramid p3(1, ca, false);
    CHECK(p3.GetPyramidImage(0).at<float>(9, 3) == -5.0f);
    MARK("p3 ok");
    Mat c = a.clone();                                // deep copy: independent
    c.at<float>(0, 0) = 77.0f;
    ImagePyramid p4(1, ca, false);
    CHECK(p4.GetPyramidImage(0).at<float>(0, 0) == ca.at<float>(0, 0) && ca.at<float>(0, 0) != 77.0f);
  }
  MARK("mirror block done");
  // lazily downloaded outputs: a textured stereo pair with disparity 20 (19 m at the KITTI baseline: inside the 30 m depth range)
  Mat L(rows, cols, PixelType), R(rows, cols, PixelType);
  for (int y = 0; y < rows; y++)
    for (int x = 0; x < cols; x++) {
      auto tex = [](int xx, int yy) { unsigned h = ((unsigned)xx * 73856093u) ^ ((unsigned)yy * 19349663u); h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; return (float)(h & 255u); };
      L.at<float>(y, x) = tex(x, y);
      R.at<float>(y, x) = tex(x + 20, y);
    }
  DepthEstimator de(8.0f, 900.0f, 15.0f, 0.1f, 30.0f, 0.01f, 28.0f, 0.995f, 50, 4, nullptr, nullptr, 386.1448f / 718.856f, 80000);
  Mat val(rows, cols, CV_8U, 0), disp(rows, cols, PixelType), dep(rows, cols, PixelType);
  MARK("stereo pair built");
  CHECK(de.ComputeDepth(L, R, val, disp, dep) == 0);
  MARK("depth done");
  DepthPyramid dp(4, dep, false);                     // takes the device copy: dep has not been downloaded yet
  const Mat& cv = val; const Mat& cd = disp; const Mat& cdep = dep;
  MARK("depth pyramid done");
  long nval = 0, nsix = 0;
  for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) { nval += cv.at<uint8_t>(y, x); nsix += (cd.at<float>(y, x) == 20.0f); }
  CHECK(nval > 500 && nsix > 500);
  CHECK(dp.GetPyramidDepth(0).at<float>(100, 100) == cdep.at<float>(100, 100));
  dep.at<float>(100, 100) = 0.5f;                     // host write after the device wrote: download first, then modify
  DepthPyramid dp2(1, dep, false);
  CHECK(dp2.GetPyramidDepth(0).at<float>(100, 100) == 0.5f && dp2.GetPyramidDepth(0).at<float>(100, 101) == cdep.at<float>(100, 101));
  MARK("all checks done");
  std::printf("OK valid=%ld\n", nval);
  std::fflush(stdout);
  return 0;
}
