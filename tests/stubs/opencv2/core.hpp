// tests/stubs/opencv2/core.hpp — TEST SCAFFOLDING ONLY: a minimal stand-in for <opencv2/core.hpp>, written from OpenCV's documented
// public API (cv::Mat: rows, cols, data, step, u, type(), total(), elemSize(), isContinuous(), ptr / at, create / clone / copyTo /
// convertTo (8U -> 32F, as the runner's load_data uses it), reference-counted header copies — the count lives in the UMatData that the
// public member `u` points to; u == nullptr for a header over user data, as in OpenCV —, ROI views, user-data headers with a row step;
// cv::Scalar, cv::Size, cv::Rect, cv::Point, cv::sum, Mat::mul and the constants the drop-in classes and the reference's callers name). This image has no OpenCV; the stub exists so that the -DODOMETRY_SHIM_WITH_OPENCV branch of
// include/odometry_shim.hpp — the one a maintainer of the reference would build — goes through a compiler and a GPU run
// (tests/test_gpu_shim.py). It is not part of the product and implements no image processing.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <cstdlib>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#define CV_8U 0
#define CV_16U 2
#define CV_32F 5
#define CV_64F 6
#define CV_8UC1 0
#define CV_32FC1 5
#define CV_64FC1 6

namespace cv {
typedef unsigned char uchar;

template <class T> struct Scalar_ {
  T val[4];
  Scalar_() { val[0] = val[1] = val[2] = val[3] = 0; }
  Scalar_(T v0) { val[0] = v0; val[1] = val[2] = val[3] = 0; }
  Scalar_(T v0, T v1, T v2 = 0, T v3 = 0) { val[0] = v0; val[1] = v1; val[2] = v2; val[3] = v3; }
  T& operator[](int i) { return val[i]; }
  const T& operator[](int i) const { return val[i]; }
};
typedef Scalar_<double> Scalar;

struct Size {
  int width, height;
  Size() : width(0), height(0) {}
  Size(int w, int h) : width(w), height(h) {}
};
struct Point {
  int x, y;
  Point() : x(0), y(0) {}
  Point(int x_, int y_) : x(x_), y(y_) {}
};
struct Rect {
  int x, y, width, height;
  Rect() : x(0), y(0), width(0), height(0) {}
  Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};
enum InterpolationFlags { INTER_NEAREST = 0, INTER_LINEAR = 1 };
enum BorderTypes { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT_101 = 4 };

struct MatStep {
  size_t p[2];
  MatStep() { p[0] = p[1] = 0; }
  operator size_t() const { return p[0]; }
  size_t operator[](int i) const { return p[i]; }
};

struct UMatData {   // (OpenCV: the allocation a Mat header refers to; only what the stand-in needs)
  int refcount;
  uchar* origdata;
};

class Mat {
 public:
  enum { AUTO_STEP = 0 };
  int flags, dims, rows, cols;
  uchar* data;
  UMatData* u;   // nullptr: the pixels are user data this header does not own
  MatStep step;

  Mat() : flags(0), dims(0), rows(0), cols(0), data(nullptr), u(nullptr), type_(0) {}
  Mat(int r, int c, int type) : Mat() { create(r, c, type); }
  Mat(int r, int c, int type, const Scalar& s) : Mat() { create(r, c, type); setTo(s); }
  // header over user data: no copy, no ownership; `step_bytes` = distance between rows (AUTO_STEP: no padding)
  Mat(int r, int c, int type, void* user, size_t step_bytes = AUTO_STEP) : Mat() {
    rows = r; cols = c; type_ = type; dims = 2; data = static_cast<uchar*>(user);
    step.p[1] = elemSize(); step.p[0] = step_bytes == AUTO_STEP ? (size_t)c * elemSize() : step_bytes;
  }
  Mat(const Mat& m) : flags(m.flags), dims(m.dims), rows(m.rows), cols(m.cols), data(m.data), u(m.u), step(m.step), type_(m.type_) {
    if (u) ++u->refcount;
  }
  Mat(const Mat& m, const Rect& roi) : Mat(m) {   // a view: shares the pixels, keeps the parent's row step
    rows = roi.height; cols = roi.width;
    data = m.data + (size_t)roi.y * m.step.p[0] + (size_t)roi.x * m.elemSize();
  }
  Mat& operator=(const Mat& m) {
    if (this != &m) {
      if (m.u) ++m.u->refcount;
      release();
      flags = m.flags; dims = m.dims; rows = m.rows; cols = m.cols; data = m.data; step = m.step; type_ = m.type_; u = m.u;
    }
    return *this;
  }
  ~Mat() { release(); }
  Mat operator()(const Rect& roi) const { return Mat(*this, roi); }

  void create(int r, int c, int type) {
    if (data && rows == r && cols == c && type_ == type) return;   // cv::Mat::create keeps a fitting buffer (whoever else refers to it)
    release();
    rows = r; cols = c; type_ = type; dims = 2;
    step.p[1] = elemSize(); step.p[0] = (size_t)c * elemSize();
    const size_t n = (size_t)r * step.p[0];
    u = new UMatData{1, static_cast<uchar*>(std::malloc(n ? n : 1))};
    data = u->origdata;
  }
  void release() {
    if (u && --u->refcount == 0) { std::free(u->origdata); delete u; }
    u = nullptr; data = nullptr; rows = cols = 0;
  }
  // the conversions the reference's callers use: imread's 8-bit image to PixelType (ref: run_odometry_kitti_offline.cpp:348,358;
  // test_disparity.cpp:144 with a scale), and the result images of save_to_vis (:443-445: float -> 8-bit, 8-bit -> 16-bit), saturating
  // and rounding to nearest as cv::saturate_cast does
  void convertTo(Mat& dst, int rtype, double alpha = 1.0, double beta = 0.0) const {
    const Mat src(*this);             // (dst may be *this: a header of the source outlives dst.create)
    dst.create(rows, cols, rtype);    // cv::Mat::create keeps a fitting buffer: a Mat refilled every frame keeps its address
    const int rows = src.rows, cols = src.cols, type_ = src.type_;
    const MatStep step = src.step;
    const uchar* const data = src.data;
    for (int y = 0; y < rows; y++) {
      const uchar* sp = data + (size_t)y * step.p[0];
      uchar* dp = dst.data + (size_t)y * dst.step.p[0];
      if (type_ == CV_8U && rtype == CV_32F && alpha == 1.0 && beta == 0.0) {   // (OpenCV's convertTo is vectorised: so is the stand-in's, whatever -O level the test builds with)
        int x = 0;
#if defined(__SSE2__)
        const __m128i z = _mm_setzero_si128();
        for (; x + 16 <= cols; x += 16) {
          const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(sp + x));
          const __m128i lo = _mm_unpacklo_epi8(v, z), hi = _mm_unpackhi_epi8(v, z);
          float* d = reinterpret_cast<float*>(dp) + x;
          _mm_storeu_ps(d, _mm_cvtepi32_ps(_mm_unpacklo_epi16(lo, z)));
          _mm_storeu_ps(d + 4, _mm_cvtepi32_ps(_mm_unpackhi_epi16(lo, z)));
          _mm_storeu_ps(d + 8, _mm_cvtepi32_ps(_mm_unpacklo_epi16(hi, z)));
          _mm_storeu_ps(d + 12, _mm_cvtepi32_ps(_mm_unpackhi_epi16(hi, z)));
        }
#endif
        for (; x < cols; x++) reinterpret_cast<float*>(dp)[x] = (float)sp[x];
      }
      else if (type_ == rtype && alpha == 1.0 && beta == 0.0) std::memmove(dp, sp, (size_t)cols * src.elemSize());
      else
        for (int x = 0; x < cols; x++) {
          const double v = alpha * src.get(sp, x) + beta;
          if (rtype == CV_32F) reinterpret_cast<float*>(dp)[x] = (float)v;
          else if (rtype == CV_64F) reinterpret_cast<double*>(dp)[x] = v;
          else {
            const double hi = rtype == CV_8U ? 255.0 : 65535.0, r = std::nearbyint(v < 0.0 ? 0.0 : v > hi ? hi : v);
            if (rtype == CV_8U) dp[x] = (uchar)r; else reinterpret_cast<uint16_t*>(dp)[x] = (uint16_t)r;
          }
        }
    }
  }
  // per-element product (ref: test_disparity.cpp:165, two 8-bit masks), saturating
  Mat mul(const Mat& o) const {
    Mat out(rows, cols, type_);
    for (int y = 0; y < rows; y++)
      for (int x = 0; x < cols; x++) {
        const double v = get(data + (size_t)y * step.p[0], x) * o.get(o.data + (size_t)y * o.step.p[0], x);
        uchar* dp = out.data + (size_t)y * out.step.p[0];
        if (type_ == CV_32F) reinterpret_cast<float*>(dp)[x] = (float)v;
        else if (type_ == CV_64F) reinterpret_cast<double*>(dp)[x] = v;
        else if (type_ == CV_16U) reinterpret_cast<uint16_t*>(dp)[x] = (uint16_t)(v > 65535.0 ? 65535.0 : v);
        else dp[x] = (uchar)(v > 255.0 ? 255.0 : v);
      }
    return out;
  }
  double get(const uchar* row, int x) const {   // (stub helper: element x of a row as double)
    return type_ == CV_32F ? (double)reinterpret_cast<const float*>(row)[x] : type_ == CV_64F ? reinterpret_cast<const double*>(row)[x]
         : type_ == CV_16U ? (double)reinterpret_cast<const uint16_t*>(row)[x] : (double)row[x];
  }
  Mat clone() const { Mat m; copyTo(m); return m; }
  void copyTo(Mat& dst) const {
    dst.create(rows, cols, type_);
    for (int y = 0; y < rows; y++) std::memcpy(dst.data + (size_t)y * dst.step.p[0], data + (size_t)y * step.p[0], (size_t)cols * elemSize());
  }
  Mat& setTo(const Scalar& s) {
    for (int y = 0; y < rows; y++) {
      uchar* row = data + (size_t)y * step.p[0];
      if (type_ == CV_8U || s[0] == 0.0) { std::memset(row, type_ == CV_8U ? (int)s[0] : 0, (size_t)cols * elemSize()); continue; }
      for (int x = 0; x < cols; x++) {
        if (type_ == CV_32F) reinterpret_cast<float*>(row)[x] = (float)s[0];
        else reinterpret_cast<double*>(row)[x] = s[0];
      }
    }
    return *this;
  }
  int type() const { return type_; }
  int depth() const { return type_; }
  int channels() const { return 1; }
  size_t elemSize() const { return type_ == CV_64F ? 8 : type_ == CV_32F ? 4 : type_ == CV_16U ? 2 : 1; }
  size_t total() const { return (size_t)rows * cols; }
  bool isContinuous() const { return rows <= 1 || step.p[0] == (size_t)cols * elemSize(); }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  template <class T> T* ptr(int y = 0) { return reinterpret_cast<T*>(data + (size_t)y * step.p[0]); }
  template <class T> const T* ptr(int y = 0) const { return reinterpret_cast<const T*>(data + (size_t)y * step.p[0]); }
  template <class T> T& at(int y, int x) { return ptr<T>(y)[x]; }
  template <class T> const T& at(int y, int x) const { return ptr<T>(y)[x]; }

 private:
  int type_;
};
// sum of all elements (ref: run_odometry_kitti_offline.cpp:235, test_disparity.cpp:85,165)
inline Scalar sum(const Mat& m) {
  double s = 0.0;
  for (int y = 0; y < m.rows; y++) { const uchar* row = m.data + (size_t)y * m.step.p[0]; for (int x = 0; x < m.cols; x++) s += m.get(row, x); }
  return Scalar(s);
}
}  // namespace cv
